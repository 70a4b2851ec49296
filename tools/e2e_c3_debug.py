"""Where the file-to-file time of a 60-minute 48 kHz capture goes (BASELINE configs[2] size: 345 MB wav, 158 MB image): the library's
own stage prints (WFX_DEBUG=1) around Demodulator.process() / save_output_image()."""
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["WFX_DEBUG"] = "1"
from wefax_amd import Demodulator, synth
x = synth.config_c3(noise=0.05, seed=0)
td = tempfile.mkdtemp(prefix="wfx_e2e_", dir="/dev/shm")
try:
    wav = os.path.join(td, "in.wav")
    synth.write_wav(wav, 48000, x)
    for k in range(3):
        t0 = time.perf_counter()
        d = Demodulator(wav, 120, quiet=True, tcp_stream=False)
        d.process()
        t1 = time.perf_counter()
        d.save_output_image(os.path.join(td, f"o{k}.png"))
        t2 = time.perf_counter()
        d.close()
        print(f"pass {k}: process {1e3 * (t1 - t0):.2f} ms, save {1e3 * (t2 - t1):.2f} ms", flush=True)
finally:
    shutil.rmtree(td, ignore_errors=True)
