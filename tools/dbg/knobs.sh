cd $GRAFT_REPO_ROOT
S="2951x235864 2951x58717 1500x235864"
run() { echo "== $1"; env $1 timeout -k 10 100 python -u tools/knn_shapes.py $S 2>&1 | grep "^Q="; }
run "X=1"
run "IMATCH_KNN_CHUNKS=40"
run "IMATCH_KNN_CHUNKS=160"
run "IMATCH_KNN_YOUNG=64"
run "IMATCH_KNN_MFMA16=0"
run "IMATCH_KNN_Q16_SMALL=0"
run "IMATCH_KNN_ROT=0"
