set -e
for o in 0 1 2; do
  echo "== IMATCH_SIFT_ORDER=$o"
  IMATCH_SIFT_ORDER=$o timeout -k 10 120 python tools/extract_profile.py 8192 4 2>&1 | grep -v amdgpu.ids
  IMATCH_SIFT_ORDER=$o timeout -k 10 120 python tools/extract_profile.py 16384 3 2>&1 | grep -v amdgpu.ids
  IMATCH_SIFT_ORDER=$o timeout -k 10 200 python tools/batch_profile.py 256 8192 6 2>&1 | grep -v amdgpu.ids | tail -9
done
