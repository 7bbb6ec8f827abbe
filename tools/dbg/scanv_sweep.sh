# development: the extremum scan with 1 / 2 / 4 columns per lane on the 8192^2 and 16384^2 maps and a 256-frame batch
set -e
for v in 1 2 4; do
  echo "== IMATCH_SIFT_SCANV=$v"
  IMATCH_SIFT_SCANV=$v timeout -k 10 120 python tools/extract_profile.py 8192 4 2>&1 | grep -v amdgpu.ids | tail -2
  IMATCH_SIFT_SCANV=$v timeout -k 10 120 python tools/extract_profile.py 16384 3 2>&1 | grep -v amdgpu.ids | tail -1
  IMATCH_SIFT_SCANV=$v timeout -k 10 200 python tools/batch_profile.py 256 8192 6 2>&1 | grep -v amdgpu.ids | tail -1
done
