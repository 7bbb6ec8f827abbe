# development: a second library with the kNN kernel's time stamps compiled in, beside the product (which is left alone):
# image_matching_amd/libimatch_trace.so; run the trace tools with IMATCH_LIB pointing at it
set -e
cd "$(dirname "$0")/../.."
IMATCH_LIB_OUT=$PWD/image_matching_amd/libimatch_trace.so IMATCH_BUILD_DEFINES="-DKNN_TRACE ${KT_FLAGS}" python -m image_matching_amd.build --force >/dev/null
echo "trace build done: IMATCH_LIB=$PWD/image_matching_amd/libimatch_trace.so python tools/dbg/knn_trace.py ..."
