set -e
R=$GRAFT_REPO_ROOT; cd $R
for V in 0 1 0 1; do
  BCE_EXTRA_FLAGS="-DBCE_AP_PREFETCH=$V" python3 openfhe-boolean-circuit-evaluator_amd/build.py --force > /dev/null 2>&1
  echo "== BCE_AP_PREFETCH=$V"; python3 tools/ap_key_locality.py 2>&1 | tail -2
done
BCE_EXTRA_FLAGS="" python3 openfhe-boolean-circuit-evaluator_amd/build.py --force > /dev/null 2>&1
