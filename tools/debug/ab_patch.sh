cd $GRAFT_REPO_ROOT
python - <<'PY'
# A/B inside the step: the knob is a process-wide debug switch, so two bench processes
PY
bash tools/ab_bench.sh "gathered 3x3|PD_IG_PATCH=0" "patch 3x3|PD_IG_PATCH=1" "gathered 3x3|PD_IG_PATCH=0" "patch 3x3|PD_IG_PATCH=1"
