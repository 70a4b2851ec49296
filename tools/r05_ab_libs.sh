#!/bin/bash
# A/B of library variants on the 60-minute stream, alternating, each in a process of its own:  bash tools/r05_ab_libs.sh "" flat nt
for rep in 1 2; do
for v in "$@"; do
  if [ -n "$v" ] && [ "$v" != "default" ]; then export WFX_LIB=$PWD/wefax_amd/variants/libwefax_hip.$v.so; else unset WFX_LIB; fi
  echo "== ${v:-default} (pass $rep)"
  WFX_LAB_SHORT=1 timeout 600 python tools/ingest_lab.py stream 2>&1 | grep -v "shader clock" | tail -4
done; done
