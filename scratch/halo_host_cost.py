import sys, time
sys.path.insert(0, '.')
import torch
from drake_amd import GpuMpm, scenes
bits, layers, res = scenes.CONFIGS['cloth_1m']
g = GpuMpm(bits)
scenes.populate(g, scenes.cloth_stack(layers, res, bits))
nb = (1 << bits) // 4
cap = 512
nbytes = g.halo_buffer_bytes(cap)
dev = torch.device('cuda', 0)
send = [torch.zeros(nbytes, dtype=torch.uint8, device=dev) for _ in range(2)]
recv = [torch.zeros(nbytes, dtype=torch.uint8, device=dev) for _ in range(2)]
stream = torch.cuda.Stream()
g.set_stream(stream.cuda_stream)
zones = [(nb // 4 - 2, nb // 4 + 1, nb // 2), (3 * nb // 4 - 2, 3 * nb // 4 + 1, -nb // 2)]
za = g.halo_zone_args(zones, [t.data_ptr() for t in send])
ra = g.halo_buffer_args([t.data_ptr() for t in recv])
dt = 1e-3
def run(n, copy):
    with torch.cuda.stream(stream):
        for _ in range(n):
            g.substep_begin_halo(dt, za, cap)
            if copy:   # stand-in for the exchange: device copies on the same stream
                recv[0].copy_(send[1], non_blocking=True)
                recv[1].copy_(send[0], non_blocking=True)
            g.substep_end_halo(dt, -1, ra, cap)
for copy in (False, True):
    run(20, copy); g.gpu_sync()
    t = time.perf_counter(); run(200, copy); t_host = time.perf_counter() - t; g.gpu_sync(); t_all = time.perf_counter() - t
    print('copy', copy, 'host enqueue us/step', t_host / 200 * 1e6, 'total us/step', t_all / 200 * 1e6, g.stats()['error_flags'])
