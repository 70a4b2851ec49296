for v in "" p2 p1; do
  if [ -n "$v" ]; then export WFX_LIB=$PWD/wefax_amd/variants/libwefax_hip.$v.so; else unset WFX_LIB; fi
  echo "== pieces variant: ${v:-3 (product)}"
  timeout 600 python tools/ingest_lab2.py 2>&1 | sed -n 3,7p
done
