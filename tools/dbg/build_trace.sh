# development: libimatch.so with the kNN kernel's time stamps compiled in (rebuild with `python -m image_matching_amd.build --force` afterwards)
set -e
cd "$(dirname "$0")/../.."
python -m image_matching_amd.build >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fno-honor-nans -DKNN_TRACE ${KT_FLAGS} -c image_matching_amd/csrc/knn.hip -o /tmp/knn_trace.o
objs=$(ls image_matching_amd/csrc/build/*.o | grep -v '/knn.o')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o image_matching_amd/libimatch.so $objs /tmp/knn_trace.o
echo trace build done
