import sys, time
sys.path.insert(0, '.')
import torch
from drake_amd import GpuMpm, scenes
bits, layers, res = scenes.CONFIGS['cloth_1m']
m = GpuMpm.default_material(); m.gravity = 0.0   # a scene that stays the same over the whole measurement
g = GpuMpm(bits, m)
scenes.populate(g, scenes.cloth_stack(layers, res, bits, vel_amp=0.0))
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
side = torch.cuda.Stream()
def run(n, copy, split):
    with torch.cuda.stream(stream):
        for _ in range(n):
            g.substep_begin_halo(dt, za, cap)
            if copy:   # stand-in for the exchange: device copies on a second stream, like RCCL's
                side.wait_stream(stream)
                with torch.cuda.stream(side):
                    recv[0].copy_(send[1], non_blocking=True)
                    recv[1].copy_(send[0], non_blocking=True)
            if split:
                g.substep_mid_halo(dt, -1)
            if copy:
                stream.wait_stream(side)
            g.substep_end_halo(dt, -1, ra, cap)
for copy, split in ((False, False), (False, True), (True, False), (True, True)):
    run(20, copy, split); g.gpu_sync()
    t = time.perf_counter(); run(200, copy, split); t_host = time.perf_counter() - t; g.gpu_sync(); t_all = time.perf_counter() - t
    print('copy', copy, 'split', split, 'host enqueue us/step', t_host / 200 * 1e6, 'total us/step', t_all / 200 * 1e6, g.stats()['error_flags'])

# the library's own chain: RCCL send/recv to self on the engine's stream, one host call per batch
g2 = GpuMpm(bits, m)
scenes.populate(g2, scenes.cloth_stack(layers, res, bits, vel_amp=0.0))
g2.chain_init(GpuMpm.chain_unique_id(), 0, 1, nb // 4, 3 * nb // 4, nb // 2, 2, cap, periodic=True)
g2.chain_substeps(20, dt, -1); g2.gpu_sync()
t = time.perf_counter(); g2.chain_substeps(200, dt, -1); t_host = time.perf_counter() - t; g2.gpu_sync(); t_all = time.perf_counter() - t
print('native RCCL ring of one: host enqueue us/step', t_host / 200 * 1e6, 'total us/step', t_all / 200 * 1e6, g2.stats()['error_flags'])
