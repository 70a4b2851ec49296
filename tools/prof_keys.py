import sys, os
sys.path.insert(0, os.getcwd())
from wefax_amd import _native as nat, synth
from wefax_amd.wefax import DecodeJob
x = synth.config_c2(noise=0.05, seed=0)
ctx = nat.Context(0)
job = DecodeJob(ctx, x, 11025, 120, hilbert_mode=nat.WFX_HILBERT_FMM)
job.run(); job.result()
ctx.profile_reset(); ctx.profile_enable(True)
for _ in range(5): job.run()
ctx.sync(); ctx.profile_enable(False)
print({k: (v[0], round(1e3*v[1]/5,1)) for k,v in ctx.profile().items()})
