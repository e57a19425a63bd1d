cd /root/repo
for V in "X=0" "TFRA_STEP_OWN_SLICE=352" "TFRA_STEP_OWN_SLICE=512" "TFRA_STEP_OWN_SLICE=576" "TFRA_STEP_OWN_SLICE=640" "TFRA_STEP_OWN_SLICE=704" "TFRA_STEP_OWN_SLICE=768"; do
  echo "== $V"; env $V python scripts/mb_owner_step.py 2>&1 | grep "distinct ids only"
done
