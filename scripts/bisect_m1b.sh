cd /root/repo
export TFRA_BENCH_DETAIL_DIR=/tmp
for V in "X=1 --slots 500000000" "TFRA_RP_IPT=1 --slots 999999992" "AMD_SERIALIZE_KERNEL=3 --slots 999999992" "X=1 --slots 800000000"; do
  E=${V%% *}; A=${V#* }
  echo "== $V"
  env $E timeout 300 python bench.py --config m1b --no-secondary --no-cpu-baseline --steps 20 --warmup 5 $A 2>&1 >/dev/null | grep -v amdgpu.ids | tail -4
done
