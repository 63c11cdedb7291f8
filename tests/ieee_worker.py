"""Runs in a process of its own with MPM_HIP_LIBRARY pointing at the -DMPM_FEM_MATH=0 build of the engine (a process
holds ONE engine library): see tests/test_ieee_variant_gpu.py.  usage: ieee_worker.py <out.npz>"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one_substep_errors():
    """(velocity error of one substep against the float oracle relative to the PLAIN max|v|, the same in units of the
    float noise of that substep) on two states: the 256^3 parity scene moving at 0.8 m/s, and config 1 as released"""
    from drake_amd import ARR as A, scenes
    from tests.helpers import build_pair, float_noise_of_a_substep
    res = {}
    for tag, dt, make in (("256^3 scene", 2e-4, lambda: build_pair(domain_bits=8, layers=3, res=40, z0=0.5, side=0.16, vel_amp=0.3)),
                          ("config 1", 1e-3, lambda: build_pair(sheets=scenes.cloth_stack(*scenes.CONFIGS["plumbing_64k"][1:], scenes.CONFIGS["plumbing_64k"][0]),
                                                                 domain_bits=scenes.CONFIGS["plumbing_64k"][0]))):
        o, g = make()
        if tag.startswith("256"):
            o.vel[:, 2] -= 0.5
            g.upload_particle_state(o.pos, o.vel, o.C, None, o.F)
        noise = float_noise_of_a_substep(o, dt)
        o.substep(dt, -1)
        for f in (g.rebuild_mapping, ):
            f(False)
        g.calc_fem_state_and_force(dt); g.particle_to_grid(dt); g.update_grid(-1); g.grid_to_particle(dt)
        g.gpu_sync()
        err = float(np.abs(g.download(A.VELOCITIES).astype(np.float64) - o.vel).max())
        eF = float(np.abs(g.download(A.DEFORMATION_GRADIENTS).astype(np.float64) - o.F).max())
        res[tag] = dict(vel_rel_plain=err / float(np.abs(o.vel).max()), vel_in_noise=err / noise, F_abs=eF,
                        max_abs_vel=float(np.abs(o.vel).max()), float_noise=noise)
        g.destroy()
    return res


def main(out):
    from drake_amd import ARR as A, GpuMpm, library_path, scenes
    assert os.environ.get("MPM_HIP_LIBRARY") and library_path() != os.environ["MPM_HIP_LIBRARY"]
    rel = one_substep_errors()
    # ---- config 2: one substep from the initial state, for the comparison with the product build -------------------
    bits, layers, res = scenes.CONFIGS["cloth_1m"]
    g = GpuMpm(bits)
    scenes.populate(g, scenes.cloth_stack(layers, res, bits))
    g.substep(1e-3, -1)
    g.gpu_sync()
    np.savez(out, vel=g.download(A.VELOCITIES), F=g.download(A.DEFORMATION_GRADIENTS), rel=json.dumps(rel))


if __name__ == "__main__":
    main(sys.argv[1])
