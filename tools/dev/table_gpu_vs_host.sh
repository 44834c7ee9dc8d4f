set -e
python tools/cli_scale.py 20000 20000 0 > /tmp/s.log 2>&1 || true
D=$(grep -o '/tmp/lx_cli_scale_[a-z0-9_]*' /tmp/s.log | head -1); if [ -z "$D" ]; then D=$(ls -dt /tmp/lx_cli_scale_* | head -1); fi
L=lambda_amd/csrc/lambda3
$L mkindexp -d $D/db.fasta -i $D/h.lba --table host 2>&1 | tail -1
$L mkindexp -d $D/db.fasta -i $D/g.lba --table gpu 2>&1 | tail -1
cmp $D/h.lba $D/g.lba && echo "index files identical (protein)"
python tools/cli_scale_nucl.py 1000 5 > /tmp/n.log 2>&1 || true
N=$(ls -dt /tmp/lx_cli_scale_nucl_* | head -1)
$L mkindexn -d $N/g.fasta -i $N/h.lba --table host 2>&1 | tail -1; $L mkindexn -d $N/g.fasta -i $N/g.lba --table gpu 2>&1 | tail -1; cmp $N/h.lba $N/g.lba && echo "index files identical (nucleotide)"
$L mkindexbs -d $N/g.fasta -i $N/hb.lba --table host 2>&1 | tail -1; $L mkindexbs -d $N/g.fasta -i $N/gb.lba --table gpu 2>&1 | tail -1; cmp $N/hb.lba $N/gb.lba && echo "index files identical (bisulfite)"
