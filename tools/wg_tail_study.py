#!/usr/bin/env python3
"""Where the ordered fill's workgroups finish, launch after launch (config 3, full size) -- the input to any static
re-balancing of the partition.  Needs the measurement build of the library, whose fill_ordered_kernel leaves three
real-time stamps (100 MHz) per workgroup: entry, end of its stream, exit.

  make -C sxmc_amd/csrc VARIANT=_stamps EXTRA=-DSXMC_WG_STAMPS=1
  SXMC_HIP_LIB=sxmc_amd/csrc/libsxmc_hip_stamps.so python3 tools/wg_tail_study.py [launches=60] [out.json]

Prints, per member of the launch (its workgroups are consecutive), the mean and spread of the workgroups' stream-end
and exit times from the launch's first entry, the part of the spread that PERSISTS from launch to launch (per-workgroup
mean) and the part that does not, and what a partition that gave every workgroup a share proportional to its
persistent speed could gain at best.  Not part of the product."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    nlaunch = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    out = sys.argv[2] if len(sys.argv) > 2 else None
    import torch
    import bench
    from sxmc_amd import capi
    lib = capi.load()
    if not hasattr(lib, "sxmc_debug_read_wg_stamps"):
        raise SystemExit("this library has no workgroup stamps: build VARIANT=_stamps and set SXMC_HIP_LIB")
    lib.sxmc_debug_read_wg_stamps.argtypes = [C.c_void_p, C.c_int]
    lib.sxmc_debug_read_wg_stamps.restype = C.c_int
    args = bench.parse_args(["--also", "none", "--experiments", "0", "--no-cpu-baseline", "--form", "fused"])
    dev = torch.device("cuda:0")
    leg = bench.Leg(args, torch, dev, "c3", args.form, False, args.seed, 12345)
    leg.setup(10, 10)
    for _ in range(50):
        leg.one_step()
    capi.synchronize()
    info = leg.m.group.LaunchInfo()
    grid = int(info.split("grid=")[1].split()[0])
    stamps = np.zeros(3 * 4096, np.uint64)
    rows = []
    for _ in range(nlaunch):
        leg.one_step()
        capi.synchronize()
        rc = lib.sxmc_debug_read_wg_stamps(stamps.ctypes.data_as(C.c_void_p), stamps.size)
        assert rc == 0
        s = stamps.reshape(3, 4096)[:, :grid].astype(np.int64)
        t0 = s[0].min()
        rows.append((s - t0) * 0.01)      # microseconds (100 MHz)
    a = np.stack(rows)                    # [launch, which, workgroup]
    nsig = leg.w.nsignals
    sizes = [s.samples.shape[0] for s in leg.w.signals]
    # the planner's apportionment (sxplan::apportion_workgroups): floor shares + largest remainders
    tot = float(sum(sizes))
    share = [grid * n / tot for n in sizes]
    K = [max(1, int(x)) for x in share]
    order = sorted(range(nsig), key=lambda j: -(share[j] - int(share[j])))
    i = 0
    while sum(K) < grid:
        K[order[i % nsig]] += 1
        i += 1
    member = np.repeat(np.arange(nsig), K)[:grid]
    print(info)
    print("launches %d, grid %d, workgroups per member %s" % (nlaunch, grid, K))
    entry, stream_end, exit_ = a[:, 0], a[:, 1], a[:, 2]
    print("entry: last workgroup enters %.2f us after the first (mean over launches)" % entry.max(axis=1).mean())
    print("exit : first %.1f  mean %.1f  last %.1f us (means over launches)" % (
        exit_.min(axis=1).mean(), exit_.mean(axis=1).mean(), exit_.max(axis=1).mean()))
    print("stream end: first %.1f  mean %.1f  last %.1f us; flush (exit - stream end) mean %.2f us" % (
        stream_end.min(axis=1).mean(), stream_end.mean(axis=1).mean(), stream_end.max(axis=1).mean(),
        (exit_ - stream_end).mean()))
    dur = stream_end - entry                       # what a share of the rows costs this workgroup
    pers = dur.mean(axis=0)                        # per workgroup, over launches
    resid = dur - pers[None, :]
    print("stream time per workgroup: mean %.1f us; persistent spread (std of per-workgroup means) %.2f us, "
          "launch-to-launch noise (std of residuals) %.2f us" % (pers.mean(), pers.std(), resid.std()))
    rec = {"launch_plan": info, "grid": grid, "K": K, "per_member": []}
    for j in range(nsig):
        sel = member == j
        p = pers[sel]
        print("member %2d: %2d workgroups  stream %.1f us (min %.1f max %.1f, std %.2f)  exit mean %.1f last %.1f" % (
            j, sel.sum(), p.mean(), p.min(), p.max(), p.std(), exit_[:, sel].mean(), exit_[:, sel].max(axis=1).mean()))
        rec["per_member"].append({"workgroups": int(sel.sum()), "stream_mean": float(p.mean()), "stream_min": float(p.min()),
                                  "stream_max": float(p.max()), "stream_std": float(p.std())})
    xcc = np.arange(grid) % 8
    print("by XCC (workgroup index mod 8): " + "  ".join("%d: %.1f" % (x, pers[xcc == x].mean()) for x in range(8)))
    # ideal static re-balancing inside each member: shares proportional to speed = 1 / persistent time, so that every
    # workgroup of a member takes the member's harmonic-mean time; the launch then ends with the slowest member + noise
    ideal_member = np.array([len(pers[member == j]) / (1.0 / pers[member == j]).sum() for j in range(nsig)])
    now_last = (entry + dur).max(axis=1).mean()
    sim = entry + ideal_member[member][None, :] + resid      # same entries, same noise, balanced persistent part
    print("stream end of the launch: now %.1f us; with shares proportional to each workgroup's persistent speed %.1f us "
          "(%.1f us = %.1f %% of the launch)" % (now_last, sim.max(axis=1).mean(), now_last - sim.max(axis=1).mean(),
                                                100 * (now_last - sim.max(axis=1).mean()) / exit_.max(axis=1).mean()))
    rec.update({"stream_end_last_now": float(now_last), "stream_end_last_balanced": float(sim.max(axis=1).mean()),
                "persistent_std": float(pers.std()), "noise_std": float(resid.std()),
                "per_workgroup_stream_us": [float(x) for x in pers]})
    # do the persistent times persist?  first half of the launches against the second
    h = nlaunch // 2
    c = np.corrcoef(dur[:h].mean(axis=0), dur[h:].mean(axis=0))[0, 1]
    print("per-workgroup mean stream time, first half of the launches against the second: correlation %.3f" % c)
    rec["half_correlation"] = float(c)
    if out:
        json.dump(rec, open(out, "w"))
    leg.close()


if __name__ == "__main__":
    main()
