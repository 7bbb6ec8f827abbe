set -e
S="1500x58717 2951x58717 800x58717 2951x235864"
echo "== default"; timeout -k 10 100 python tools/knn_shapes.py $S
for c in 24 40 56 80; do echo "== CHUNKS=$c"; IMATCH_KNN_CHUNKS=$c timeout -k 10 100 python tools/knn_shapes.py $S; done
