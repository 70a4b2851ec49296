import cProfile, pstats, os, sys, tempfile, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wefax_amd import Demodulator, synth
x = synth.synth_capture(11025.0, noise=0.05, seed=0, image_lines=1100)
with tempfile.TemporaryDirectory() as td:
    wav, png = os.path.join(td, "in.wav"), os.path.join(td, "out.png")
    synth.write_wav(wav, 11025, x)
    d = Demodulator(wav, 120, quiet=True, tcp_stream=False); d.process(); d.save_output_image(png); d.close()
    import time; t0 = time.perf_counter()
    d = Demodulator(wav, 120, quiet=True, tcp_stream=False, device=0); d.process(); t1 = time.perf_counter(); d.save_output_image(png + '2.png'); print('plain timing: process %.2f ms, save %.2f ms' % ((t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3)); d.close()
    pr = cProfile.Profile(); pr.enable()
    d = Demodulator(wav, 120, quiet=True, tcp_stream=False); d.process(); d.save_output_image(png + '3.png')
    pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[:3500])
