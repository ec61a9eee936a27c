for c in 8388608 2097152 1048576 786432 524288 262144; do
  echo "== NFE_BWD_CHUNK=$c"; NFE_BWD_CHUNK=$c python3 tools/time_backward.py 4 128 48 48 256 2>&1 | grep -v "^/opt" | tail -1
done
