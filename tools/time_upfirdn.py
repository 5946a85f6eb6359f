"""Per-output and shared-window polyphase kernels side by side through the C ABI:
python tools/time_upfirdn.py  (rows of 30 s at the source rate, 128 rows; ms per call)."""
import torch

from mlx_audio_amd import _lib
from mlx_audio_amd.primitives.resample import _kaiser_lowpass, _phase_table_dev, _poly_layout

CASES = [(44100, 16000), (48000, 16000), (16000, 48000), (32000, 16000), (8000, 16000),
         (44100, 48000), (22050, 16000), (24000, 16000), (96000, 16000), (44100, 22050),
         (48000, 44100), (16000, 44100), (64000, 16000)]


def main():
    import math
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    rows = 128
    stream = torch.cuda.current_stream(dev).cuda_stream
    for src, dst in CASES:
        g = math.gcd(src, dst)
        up, down = dst // g, src // g
        n_in = src * 30
        x = torch.randn((rows, n_in), device=dev)
        h, half_len = _kaiser_lowpass(up, down)
        pre, post, skip, n_out = _poly_layout(n_in, up, down, half_len, len(h))
        P = -(-(pre + len(h) + post) // up)
        margin = lib.mia_upfirdn_margin(up, down, P, n_out)
        out = torch.empty((rows, n_out), device=dev)
        plain = _phase_table_dev(up, down, pre, post, dev)

        def run_plain():
            _lib.check(lib.mia_upfirdn(x.data_ptr(), rows, n_in, plain.data_ptr(), up, down, P,
                                       skip, n_out, 0, out.data_ptr(), stream))

        def timed(fn):
            for _ in range(3):
                fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                fn()
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / 10

        t_plain = timed(run_plain)
        ref = out.clone()
        t_shared = float("nan")
        same = None
        if margin:
            padded = _phase_table_dev(up, down, pre, post, dev, margin)

            def run_shared():
                _lib.check(lib.mia_upfirdn_shared(x.data_ptr(), rows, n_in, padded.data_ptr(),
                                                  margin, up, down, P, skip, n_out, 0,
                                                  out.data_ptr(), stream))
            out.zero_()
            t_shared = timed(run_shared)
            same = bool(torch.equal(out, ref))
        gb = 4 * rows * (n_in + n_out) / 1e9
        print(f"{src:6d} -> {dst:6d}  up {up:4d} down {down:4d} taps {P:4d}  per-output {t_plain:7.3f} ms"
              f"  shared {t_shared:7.3f} ms  identical {same}  ({gb:.2f} GB)", flush=True)


if __name__ == "__main__":
    main()
