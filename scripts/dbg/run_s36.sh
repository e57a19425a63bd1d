cd /root/repo
timeout 900 python scripts/mb_overlap.py --skip-old --depth 4 --variants 2,5,4,258,261 --out gpurun_out/s36_mb.json > gpurun_out/s36_mb.log 2>&1
timeout 900 python scripts/mb_overlap.py --skip-old --depth 1 --variants 277 --out gpurun_out/s36t_mb.json > gpurun_out/s36t_mb.log 2>&1
