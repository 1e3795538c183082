"""Probe: does a B <= 2048 inference forward gain from running two half-batches on two HIP streams?  (The recurrent
kernels launch one 16-row tile per workgroup: B = 1024 -> 128 workgroups on 256 CUs; their step time is the serial
chain, so a second stream's GEMM / row-wise kernels can use the idle CUs.)  Prints ms per forward for one stream / two."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ode_bci_amd import EnhancedLSTMModel          # noqa: E402
from lstm_ode_bci_amd import synthetic as syn           # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    H = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    m = EnhancedLSTMModel(61, H, 3, 2, 0.4, True).to(dev).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in syn.make_state_dict(61, H, 3, 2, True).items()})
    side = torch.cuda.Stream(device=dev)
    for B in (256, 512, 1024, 2048, 4096):
        x, _ = syn.make_windows(B, 256, 61, seed=3)
        x = torch.from_numpy(x).to(dev)
        for amp in (False, True):
            def one():
                with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                    return m(x)

            def two(parts=2):
                main = torch.cuda.current_stream()
                h = (B // parts + 31) // 32 * 32
                ev = torch.cuda.Event()
                ev.record(main)
                with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                    with torch.cuda.stream(side):
                        side.wait_event(ev)
                        b = m(x[h:])
                    a = m(x[:h])
                main.wait_stream(side)
                b.record_stream(main)
                return torch.cat([a, b])

            ref = one()
            got = two()
            same = bool(torch.equal(ref, got))
            res = {}
            for name, fn in (("one", one), ("two", two)):
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n = 10
                for _ in range(n):
                    fn()
                torch.cuda.synchronize()
                res[name] = (time.perf_counter() - t0) / n * 1e3
            print(f"H={H} B={B:5d} amp={int(amp)}  one stream {res['one']:7.3f} ms   two streams {res['two']:7.3f} ms   "
                  f"x{res['one'] / res['two']:.2f}   bit-equal {same}", flush=True)


if __name__ == "__main__":
    main()
