"""Does pass 1 of one block beside pass 2 of another move more bytes per second than the two one after the other?
Two SpectrumSinks on two streams of the same GPU, each cycling its own resident blocks (C3 shape), against one."""
import sys, time
sys.path.insert(0, ".")
import torch
from webradio_amd.device import Device, Spectrum

N, FFT, HOP = 4_000_000, 65536, 32768
rows = (N - FFT) // HOP + 1
nb = 6
g = torch.Generator(device="cuda").manual_seed(1)


def mk():
    return [torch.randn(2 * N, device="cuda", generator=g) * 0.1 for _ in range(nb)], \
           [torch.empty(rows * FFT, device="cuda") for _ in range(nb)]


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
d1, d2 = Device(0, s1.cuda_stream), Device(0, s2.cuda_stream)
b1, o1 = mk()
b2, o2 = mk()
sp1, sp2 = Spectrum(d1, FFT, HOP), Spectrum(d2, FFT, HOP)
torch.cuda.synchronize()


def run(two, steps=120):
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            sp1.batch_db(b1[i % nb], rows, o1[i % nb])
            if two:
                sp2.batch_db(b2[i % nb], rows, o2[i % nb])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    n = steps * (2 if two else 1)
    return dt / n * 1e6


print("one stream : %.1f us per 121-frame block" % run(False))
print("two streams: %.1f us per 121-frame block (both streams' blocks counted)" % run(True))
print("one stream : %.1f us per 121-frame block" % run(False))
