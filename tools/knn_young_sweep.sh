# kNN bound-exchange policy sweep (IMATCH_KNN_YOUNG = stages with an exchange on every stage, IMATCH_KNN_YOUNG_WG = first
# workgroups it applies to); run on the GPU box: bash tools/knn_young_sweep.sh "<young> <wg>" ...
cd /tmp
R=$GRAFT_REPO_ROOT
for cfg in "$@"; do
  set -- $cfg
  echo "young $1 wg $2"
  IMATCH_KNN_YOUNG=$1 IMATCH_KNN_YOUNG_WG=$2 timeout -k 10 100 python $R/tools/knn_bench.py 188136 235864 10 2>/dev/null || exit 1
  IMATCH_KNN_YOUNG=$1 IMATCH_KNN_YOUNG_WG=$2 timeout -k 10 100 python $R/tools/knn_bench.py 94068 235864 10 2>/dev/null || exit 1
done
