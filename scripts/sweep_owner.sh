cd /root/repo
for V in "X=0" "TFRA_STEP_VARIANT=2"; do
  echo "== $V"; env $V python scripts/mb_owner_step.py 2>&1 | grep "distinct ids only\|full batches"
done
