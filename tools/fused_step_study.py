#!/usr/bin/env python3
"""Where the time goes inside the fused step kernel (fill_ordered_step_kernel, config 3, full size): the real-time stamps
(100 MHz) of the measurement build -- every fill workgroup's entry / end of stream / exit, every role workgroup's
entry / sight of the fill's end / exit -- over a number of launches.

  make -C sxmc_amd/csrc VARIANT=_stamps EXTRA=-DSXMC_WG_STAMPS=1
  SXMC_HIP_LIB=sxmc_amd/csrc/libsxmc_hip_stamps.so python3 tools/fused_step_study.py [launches=60]
Not part of the product."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    nlaunch = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    import torch
    import bench
    from sxmc_amd import capi
    lib = capi.load()
    if not hasattr(lib, "sxmc_debug_read_wg_stamps"):
        raise SystemExit("this library has no workgroup stamps: build VARIANT=_stamps and set SXMC_HIP_LIB")
    lib.sxmc_debug_read_wg_stamps.argtypes = [C.c_void_p, C.c_int]
    lib.sxmc_debug_read_wg_stamps.restype = C.c_int
    args = bench.parse_args(["--also", "none", "--experiments", "0", "--no-cpu-baseline"])
    leg = bench.Leg(args, torch, torch.device("cuda:0"), "c3", "graph", False, args.seed, 12345)
    leg.setup(10, 10)
    m = leg.m
    m.group.SetFusedStep(True)
    for _ in range(50):
        m.step()
    capi.synchronize()
    info = m.group.LaunchInfo()
    nfill = int(info.split("grid=")[1].split()[0])
    assert m.group.LastStepLaunches() == 1, "the step is not fused: %s" % info
    stamps = np.zeros(3 * 4096, np.uint64)
    rows = []
    for _ in range(nlaunch):
        m.step()
        capi.synchronize()
        assert lib.sxmc_debug_read_wg_stamps(stamps.ctypes.data_as(C.c_void_p), stamps.size) == 0
        s = stamps.reshape(3, 4096).astype(np.int64)
        t0 = s[0, :nfill].min()
        rows.append((s - t0) * 0.01)
        m.flush()
    a = np.stack(rows)                          # [launch, which, block] in microseconds from the first fill entry
    nroles = int((a[0, 0, nfill:nfill + 200] > 0).sum())
    fill, role = a[:, :, :nfill], a[:, :, nfill:nfill + nroles]
    mean = lambda x: float(np.mean(x))
    print(info.strip())
    print("launches %d, fill workgroups %d, role workgroups %d (finisher + %d workers)" % (nlaunch, nfill, nroles, nroles - 1))
    print("fill : exit first %.1f  mean %.1f  LAST %.1f us" % (mean(fill[:, 2].min(axis=1)), mean(fill[:, 2].mean(axis=1)),
                                                              mean(fill[:, 2].max(axis=1))))
    print("roles: entry first %.1f  finisher %.1f  last %.1f us" % (mean(role[:, 0].min(axis=1)), mean(role[:, 0, 0]),
                                                                   mean(role[:, 0].max(axis=1))))
    print("roles: see the fill's end first %.1f  finisher %.1f  last %.1f us  (after the last fill exit: %.2f .. %.2f us)" % (
        mean(role[:, 1].min(axis=1)), mean(role[:, 1, 0]), mean(role[:, 1].max(axis=1)),
        mean(role[:, 1].min(axis=1) - fill[:, 2].max(axis=1)), mean(role[:, 1].max(axis=1) - fill[:, 2].max(axis=1))))
    print("roles: exit finisher %.1f  workers first %.1f  last %.1f us" % (mean(role[:, 2, 0]), mean(role[:, 2, 1:].min(axis=1)),
                                                                          mean(role[:, 2, 1:].max(axis=1))))
    print("the launch after its last fill workgroup has left: %.2f us (finisher %.2f, last worker %.2f)" % (
        mean(role[:, 2].max(axis=1) - fill[:, 2].max(axis=1)), mean(role[:, 2, 0] - fill[:, 2].max(axis=1)),
        mean(role[:, 2, 1:].max(axis=1) - fill[:, 2].max(axis=1))))
    leg.close()


if __name__ == "__main__":
    main()
