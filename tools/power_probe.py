"""Diagnostic: is the fp32 GEMM power limited?  Same launch on random and on zero operands, with rocm-smi polled meanwhile.

    python tools/power_probe.py
"""
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from lime_cikm25_amd import ops  # noqa: E402


def poll(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--csv'], capture_output=True, text=True, timeout=5)
            out.append(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr.strip()[:200])
        except Exception as e:                                              # noqa: BLE001
            out.append('rocm-smi failed: %r' % (e,))
            return
        time.sleep(0.2)


def main():
    dev = 'cuda'
    g = torch.Generator().manual_seed(0)
    tok, E, F = 225280, 300, 512
    rnd = lambda *s: ((torch.rand(*s, generator=g) * 2 - 1)).to(dev)
    cases = {'random': (rnd(tok, E), rnd(F, E) * 0.06), 'zero A': (torch.zeros(tok, E, device=dev), rnd(F, E) * 0.06),
             'zeros': (torch.zeros(tok, E, device=dev), torch.zeros(F, E, device=dev))}
    b = rnd(F)
    out = torch.empty(tok, F, device=dev)
    r = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--csv'], capture_output=True, text=True)
    print('rocm-smi header:', (r.stdout.strip().splitlines() or [r.stderr.strip()[:200]])[0])
    for name, (a, w) in cases.items():
        for _ in range(5):
            ops.linear(a, w, b, act='relu', out=out)
        torch.cuda.synchronize()
        stop, log = threading.Event(), []
        th = threading.Thread(target=poll, args=(stop, log))
        th.start()
        n = 3000
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            ops.linear(a, w, b, act='relu', out=out)
        e1.record()
        torch.cuda.synchronize()
        stop.set()
        th.join()
        us = e0.elapsed_time(e1) * 1e3 / n
        print('%-8s %.1f us  %.1f TFLOP/s' % (name, us, 2.0 * tok * E * F / us / 1e6))
        for line in log[1:6]:
            print('    ', line)


if __name__ == '__main__':
    main()
