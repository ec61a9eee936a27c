V=nerffaceediting_amd/csrc/build/variants
for name in shipped abl2 abl6 abm26 abm30 abm62 abm94; do
  lib=""; [ "$name" != shipped ] && lib=$V/$name.so
  echo "== $name"; NFE_RENDER_LIB=$lib python3 tools/time_conv.py bf16 8 2>&1 | grep -v "^/opt" | head -2
done
