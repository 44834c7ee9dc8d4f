#!/bin/bash
# A/B of two source states on ONE box (development aid): tools/dev/ab_commits.sh "<bench args>" <patch-file-to-reverse>
# the tree as it is = B; the patch reversed = A.  (the GPU box has hipcc; .git does not travel, so A is made with patch -R)
ARGS="$1"; P="$2"
run() { for i in 1 2 3; do python bench.py $ARGS --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}; print('   ', d['value'], d.get('ms_min'), d.get('ms_max'), 'sweep ms', r.get('kernel_ms_per_call') or r.get('kernel_ms_per_launch'), 'backtrace ms', r.get('backtrace_ms_per_call'))"; done; }
python -c "import lambda_amd.build as b; b.build_product()" >/dev/null 2>&1; echo "B (tree):"; run
patch -R -p1 < $P >/dev/null && python -c "import lambda_amd.build as b; b.build_product()" >/dev/null 2>&1 && echo "A (patch reversed):" && run
patch -p1 < $P >/dev/null; python -c "import lambda_amd.build as b; b.build_product()" >/dev/null 2>&1; echo "B again:"; run
