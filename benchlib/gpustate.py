"""Clocks, power and temperature of the GPU a context runs on (amdgpu sysfs, matched by PCI bus id)."""
from __future__ import annotations

import json
import os
import subprocess
import sys
import time

import benchlib
from benchlib import REPO, HBM_PEAK_GBS, IQ_FS


# ---- what the box was doing: clocks, power, temperature from sysfs ------------------------------------------------
class GpuState:
    """Reads the amdgpu sysfs nodes of the first GPU (shader / memory / fabric clock, socket power, temperatures) -- at a
    point in time (``read``) or sampled by a thread while a timed region runs (``with GpuState.sample() as s``): a reader of the
    line can then tell a slow box from a slow kernel.  Everything is optional: a node that is missing or unreadable is left out."""
    _dev = None
    _pci = None          # PCI address of the GPU the bench's context runs on (GpuState.bind): the card to read
    _matched = False

    @classmethod
    def bind(cls, ctx):
        """Read the card whose PCI address is the context's device's (round-5 verdict: on an 8-GPU box the first card is somebody else's GPU)."""
        try:
            cls._pci = ctx.pci_bus_id().lower()
        except Exception:       # noqa: BLE001
            cls._pci = None
        cls._dev = None

    @classmethod
    def dev(cls):
        if cls._dev is None:
            import glob
            cls._dev = ""
            first = ""
            for d in sorted(glob.glob("/sys/class/drm/card*/device")):
                try:
                    if open(os.path.join(d, "vendor")).read().strip() != "0x1002" or not os.path.exists(os.path.join(d, "pp_dpm_sclk")):
                        continue
                except OSError:
                    continue
                first = first or d
                if cls._pci and os.path.basename(os.path.realpath(d)).lower() == cls._pci:
                    cls._dev, cls._matched = d, True
                    break
            if not cls._dev:
                cls._dev, cls._matched = first, False
        return cls._dev

    @staticmethod
    def _cur_mhz(path):
        try:
            for ln in open(path).read().splitlines():
                if ln.rstrip().endswith("*"):
                    return int("".join(ch for ch in ln.split(":")[1] if ch.isdigit()))
        except (OSError, ValueError, IndexError):
            pass
        return None

    @classmethod
    def read(cls) -> dict:
        import glob
        d = cls.dev()
        out = {}
        if not d:
            return out
        out["pci"] = os.path.basename(os.path.realpath(d))
        out["is_the_contexts_gpu"] = bool(cls._matched)
        for key, node in (("sclk_mhz", "pp_dpm_sclk"), ("mclk_mhz", "pp_dpm_mclk"), ("fclk_mhz", "pp_dpm_fclk")):
            v = cls._cur_mhz(os.path.join(d, node))
            if v is not None:
                out[key] = v
        for hw in glob.glob(os.path.join(d, "hwmon", "hwmon*")):
            for key, node, scale in (("power_w", "power1_average", 1e-6), ("power_w", "power1_input", 1e-6), ("temp_c", "temp1_input", 1e-3),
                                     ("temp_mem_c", "temp3_input", 1e-3), ("power_cap_w", "power1_cap", 1e-6)):
                if key in out:
                    continue
                try:
                    out[key] = round(int(open(os.path.join(hw, node)).read().strip()) * scale, 1)
                except (OSError, ValueError):
                    pass
        try:
            out["busy_pct"] = int(open(os.path.join(d, "gpu_busy_percent")).read().strip())
        except (OSError, ValueError):
            pass
        return out

    class _Sampler:
        def __init__(self, period):
            import threading
            self.period, self.rows, self._stop = period, [], threading.Event()
            self._thr = threading.Thread(target=self._run, daemon=True)

        def _run(self):
            while not self._stop.is_set():
                r = GpuState.read()
                if r:
                    self.rows.append(r)
                self._stop.wait(self.period)

        def __enter__(self):
            self._thr.start()
            return self

        def __exit__(self, *exc):
            self._stop.set()
            self._thr.join(timeout=2.0)

        def summary(self) -> dict:
            out = {"samples": len(self.rows)}
            for key in ("sclk_mhz", "mclk_mhz", "fclk_mhz", "power_w", "temp_c", "temp_mem_c"):
                vals = sorted(r[key] for r in self.rows if key in r)
                if vals:
                    out[key] = {"min": vals[0], "median": vals[len(vals) // 2], "max": vals[-1]}
            return out

    @classmethod
    def sample(cls, period: float = 0.02):
        return cls._Sampler(period)
