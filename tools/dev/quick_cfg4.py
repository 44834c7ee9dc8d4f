import sys; sys.path.insert(0,'.')
import subprocess
for lq,extra in ((200,[]),(150,[])):
    out=subprocess.run([sys.executable,'bench.py','--steps','3','--warmup','1','--no-cpu-baseline','--lq',str(lq),'--queries','60000' if lq==200 else '100000']+extra,capture_output=True,text=True).stdout.strip().splitlines()[-1]
    import json; d=json.loads(out); print(lq, d['value'], d['ms_per_step'], d['phase_ms_last_step'], d['roofline']['kernel'], [r['kernel'] for r in d['roofline_other']])
