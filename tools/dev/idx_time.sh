# development aid: wall clock of `search -i INDEX` against `search -d FASTA` on the files of tools/cli_scale.py
python tools/cli_scale.py 100000 100000 0 --via-index > /tmp/scale.log 2>&1 || true
D=$(grep -o '/tmp/lx_cli_scale_[a-z0-9_]*' /tmp/scale.log | head -1)
L=lambda_amd/csrc/lambda3
for i in 1 2; do s=$(date +%s.%N); $L searchp -q $D/q.fasta -i $D/db.lba -o $D/o1.m8 2>&1 | grep 'times'; e=$(date +%s.%N); python3 -c "print(\"wall %.2f s (index)\" % ($e - $s))"; done
for i in 1 2; do s=$(date +%s.%N); $L searchp -q $D/q.fasta -d $D/db.fasta -o $D/o2.m8 2>&1 | grep 'times'; e=$(date +%s.%N); python3 -c "print(\"wall %.2f s (fasta)\" % ($e - $s))"; done
