"""What `bench.py` is made of (round 6: split out of one 1 400-line file): `ranks` (one process per GPU: spawning, the environment's ranks, the RCCL
probe, the guard that saves the headline when a sharded part hangs), `gpustate` (clocks / power / temperature of the context's GPU from sysfs),
`measure` (HIP-event profiles -> kernel tables and the roofline object), `workloads` (one function per BASELINE config and companion), `e2e`
(wav on tmpfs -> png on tmpfs).  `bench.py` keeps the command line, the headline's JSON line and -- alone -- every use of `oracle/` (the CPU
baseline and the in-run checks reach it through `benchlib.ORACLE` / `benchlib.CPU_BASELINE`, which bench.py sets)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
IQ_FS = 1536000

ORACLE = None                  # bench.py: a callable returning the oracle module (the checker; never the thing measured)
CPU_BASELINE = None            # bench.py: cpu_baseline(x, sample_rate, lpm, faithful, what)
