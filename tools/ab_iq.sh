#!/bin/bash
# A/B of library variants on the 60-minute IQ stream inside one gpurun call:  bash tools/ab_iq.sh "" pf0 pf0p0 ...
V=$PWD/wefax_amd/variants
for rep in 1 2; do
for name in "$@"; do
    if [ -n "$name" ] && [ "$name" != "default" ]; then export WFX_LIB=$V/libwefax_hip.$name.so; else unset WFX_LIB; fi
    python bench.py --workload iq --no-cpu $AB_ARGS | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('%-10s' % '${name:-default}', d['ms_per_step'], 'ingest', k['polyphase_ingest']['us_per_step'], 'stages', k['polyphase_stages']['us_per_step'], 'fwd', k['fft_pass_fwd']['us_per_step'])"
done; done
