cd /root/repo
timeout 900 python scripts/mb_overlap.py --skip-old --depth 4 --variants 0,2,3,128,130,131 --out gpurun_out/s33_mb.json > gpurun_out/s33_mb.log 2>&1
TFRA_X=1 timeout 300 python scripts/mb_overlap.py --skip-old --depth 1 --variants 16,18,146 --out gpurun_out/s33t_mb.json > gpurun_out/s33t_mb.log 2>&1
tail -5 gpurun_out/s33_mb.log
