# whole search at several batch sizes (tools/small_batch_search.py); A/B: LIBS="ab/lib_base.so ab/lib_tiled.so" bash tools/search_batch_sizes.sh
for rep in 1 2; do
for lib in ${LIBS:-viquae_amd/csrc/libmeerqat_hip.so}; do
  for nq in ${NQS:-256 1024 4096}; do
    echo -n "$lib "
    MEERQAT_HIP_LIB=$PWD/$lib timeout 300 python3 tools/small_batch_search.py $nq 30 2>&1 | tail -1
  done
done
done
