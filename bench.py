#!/usr/bin/env python3
"""bench.py -- the hot path of NeRF-PRV on MI355X: render + score candidate views.

One "step" = one scoring round of the planner over this rank's shard of the candidate set:
march + render every view (800x800, 128 samples/ray) of the synthetic 256^3 hash-grid field,
reduce each to a PSNR/coverage score against reference images already resident in HBM,
all-gather the 16-byte records over RCCL (N > 1), rank.  BASELINE.json configs[1].

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU)

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

SEED_A, SEED_B = 0x5EED0001, 0x5EED0002
BYTES_PER_SAMPLE = 512  # L*8 corners*F*2 B = 8*8*4*2: hash-table gather per field evaluation
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def measured_traffic(args, samples_per_launch):
    """HBM-side bytes per render_queue launch from the committed PMC pass (profiles/), scaled by the
    evaluated-sample count; None unless the workload is the one that was profiled"""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if not os.path.exists(path) or (args.width, args.height, args.samples, args.field) != (800, 800, 128, "256"):
        return None
    with open(path) as f:
        k = json.load(f)["render_queue_kernel"]
    per_sample = (k["fetch_kib_per_launch"] + k["write_kib_per_launch"]) * 1024.0 / k["samples_evaluated_per_launch"]
    return per_sample * samples_per_launch


def usable_cores():
    """host cores this process may actually use: the affinity mask, capped by a cgroup CPU quota when there is one
    (the GPU boxes expose 256 logical CPUs behind a 16-CPU quota: 256 threads there are 16 cores' worth of time)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = min(n, max(1, -(-q // per)))
        except (OSError, ValueError):
            pass
    return max(1, n)


def cpu_baseline(args, tms, scale, offset, fov_x):
    """the oracle (a scalar C port of the same algorithm) on the host cores, bounded sample"""
    from oracle import oracle as orc

    threads = usable_cores()
    f = orc.OracleField(orc.desc(), seed=SEED_A)
    cams = orc.cameras_from_transforms(tms, fov_x, args.width, args.height, scale, offset)
    threads = min(threads, 256)  # the oracle's pthread pool tops out there
    band = min(args.height, max(64, threads))  # one row per thread at least: every core the line claims has work
    rows = ((args.height - band) // 2, (args.height - band) // 2 + band)
    f.render(cams[0], args.width, args.height, args.samples, 1, 1e-4, threads=threads, rows=(rows[0], rows[0] + 1))
    n_eval, n_views, t0 = 0, 0, time.perf_counter()
    for v in range(len(cams)):
        _, ne = f.render(cams[v], args.width, args.height, args.samples, 1, 1e-4, threads=threads, rows=rows)
        n_eval += ne
        n_views += 1
        if time.perf_counter() - t0 > args.cpu_seconds:
            break
    dt = time.perf_counter() - t0
    return {
        "value": n_eval / dt,
        "unit": "ray-samples/s",
        "cores": min(threads, band),
        "kind": "port",
        "sample": f"rows {rows[0]}-{rows[1]} of {n_views} views at {args.width}x{args.height}, "
                  f"{args.samples} samples/ray, {n_eval} samples evaluated in {dt:.1f} s "
                  f"(oracle/prv_oracle.c, pthreads over rows)",
        "first_hit_rays_per_s": first_hit_cpu(f, cams[0], args),
    }


def first_hit_cpu(f, cam, args):
    """the oracle's scalar first-hit DDA on ONE core, 64 rows of one view (the reference spawns one OS
    thread per voxel, main.cpp:124-130; a single tight loop is the kinder comparison)"""
    from oracle import oracle as orc

    rows = (args.height // 2 - 32, args.height // 2 + 32)
    t0 = time.perf_counter()
    orc.first_hit_image(f, cam, args.width, args.height, rows=rows)
    return (rows[1] - rows[0]) * args.width / (time.perf_counter() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--views-per-gpu", type=int, default=64)
    ap.add_argument("--width", type=int, default=800)
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--field", choices=["256", "512"], default="256")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-training", action="store_true")
    ap.add_argument("--train-steps", type=int, default=500)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("PRV_FORCE_DIST") == "1"  # the latter: exercise RCCL with one rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from nerf_prv_amd import api, planner

    ctx = api.Context(local_rank)  # raises when libprv_hip.so / the GPU is missing: no fallback
    fdict = dict(api.FIELD_256 if args.field == "256" else api.FIELD_512)
    desc = api.L.FieldDesc(**fdict)
    ctx.synthetic_model(0, desc, SEED_A)  # the field being scored
    ctx.synthetic_model(1, desc, SEED_B)  # the field the reference images come from

    # candidate set: generated hemisphere, radius 0.3 around the origin (+1e-10), object size 0.1
    # -> cameras at 1.5 cube units from the centre of the unit cube
    n_views = args.views_per_gpu * world
    pts = planner.hemisphere_generate(n_views)
    fov_x = 2.0 * np.arctan(0.5 * 1280 / 915.60668945312500)  # the reference camera's 69.9 deg
    tms, scale, offset = planner.hemisphere_transforms(pts, 0.3, 0.1, [1e-10] * 3)
    cams = ctx.cameras_from_matrices(tms, fov_x, args.width, args.height, scale, offset)
    my_ids, per_rank = planner.shard_views(n_views, rank, world, interleaved=True)  # balances pole vs equator views
    opts = api.render_opts(args.width, args.height, args.samples, 1, 1e-4)

    # reference images of this rank's views, resident in HBM before the timed region
    gt, _ = ctx.render(1, cams, my_ids, opts, want_stats=False)
    rec_dev = torch.zeros(per_rank * 16, dtype=torch.uint8, device=device)

    def step(want_stats=False):
        _, st = ctx.score_views(api.L.SCORE_PSNR_COVERAGE, [0], cams, my_ids, opts, gt=gt, records_dev=rec_dev,
                                to_host=False, want_stats=want_stats)
        records = planner.gather_records(rec_dev, per_rank, n_views, device=device, interleaved=True)  # the ONE collective
        order = api.rank_host(records, np.arange(n_views, dtype=np.int32))
        return st, records, order

    st, records, order = step(want_stats=True)  # also sizes every workspace
    evaluated_per_step, nominal_per_step, rays_per_step = st.samples_evaluated, st.samples_nominal, st.rays
    for _ in range(max(0, args.warmup - 1)):
        step()

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    ctx.profile_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        _, records, order = step()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = ctx.profile_end()

    # BASELINE config 1 analogue (the reference's CPU render path, main.cpp:98-284): first occupied voxel
    # per ray over the same views; GPU (prv_first_hit) here, the oracle's scalar DDA in cpu_baseline
    first_hit = None
    if rank == 0:
        cells = ctx.first_hit(0, cams, my_ids, args.width, args.height)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            cells = ctx.first_hit(0, cams, my_ids, args.width, args.height)
        torch.cuda.synchronize()
        dt_fh = (time.perf_counter() - t1) / 5
        first_hit = {"gpu_rays_per_s": len(my_ids) * args.width * args.height / dt_fh,
                     "hit_fraction": float((cells >= 0).float().mean().item())}

    # the other half of an NBV iteration: in-process training of the field (run.py:185-208), outside the timed
    # region and not part of `value`: a fresh 256^3 field trained on this rank's reference images
    training = None
    if rank == 0 and not args.no_training:
        tcams = ctx.cameras_from_matrices(np.asarray(tms)[my_ids], fov_x, args.width, args.height, scale, offset)
        u8, _ = ctx.render_rgba8(1, tcams, None, api.render_opts(args.width, args.height, args.samples, 1, 1e-4,
                                                                background=(0, 0, 0, 0)))
        tdesc = api.L.FieldDesc(**dict(fdict, table_amp=1e-4, density_bias=0.0))
        ctx.fresh_model(2, tdesc, 0x1234)
        tr = api.Trainer(ctx, 2, tcams, u8, api.train_opts())
        tr.steps(300)  # past the all-occupied start
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        losses = tr.steps(args.train_steps)
        torch.cuda.synchronize()
        dt_tr = time.perf_counter() - t2
        training = {"steps_per_s": args.train_steps / dt_tr, "ms_per_step": dt_tr / args.train_steps * 1e3,
                    "rays_per_step": int(tr.opts.n_rays), "samples_per_ray": int(tr.opts.n_samples),
                    "used_samples_last_batch": tr.info()["samples_last"], "loss_last": float(losses[-1]),
                    "note": "fresh field, 300 warm-up steps untimed; f16-MFMA forward, f32-MFMA backward, sparse Adam"}
        tr.close()
        tcams.close()

    tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
    tot = torch.tensor([float(evaluated_per_step), float(nominal_per_step), float(rays_per_step)], dtype=torch.float64,
                       device=device)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    elapsed = float(tmax.item())
    ev_all, nom_all, rays_all = (float(x) for x in tot.tolist())

    if rank == 0:
        k = args.steps
        launches = max(1, prof["render_launches"])
        kernel_s = prof["render_ms"] * 1e-3 / launches  # average render_queue launch duration
        samples_per_launch = evaluated_per_step * k / launches
        achieved = samples_per_launch * BYTES_PER_SAMPLE / kernel_s / 1e9
        out = {
            "metric": "ray-samples/s (field evaluations composited; candidate views rendered + scored)",
            "value": ev_all * k / elapsed,
            "unit": "ray-samples/s",
            "n_gpus": world,
            "steps": k,
            "warmup": args.warmup,
            "ms_per_step": elapsed / k * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16",
            "dtype_note": "fp16 table, blend and MLP operands (MFMA f16 -> f32 accumulate); f32 rays, positions, compositing",
            "data": "synthetic",
            "config": {
                "workload": f"render+score {args.views_per_gpu} hemisphere views/GPU, {args.width}x{args.height}, "
                            f"{args.samples} samples/ray, synthetic {args.field}^3 hash-grid field "
                            f"(L={fdict['n_levels']} F={fdict['n_features']} log2T={fdict['log2_hashmap']}), "
                            "PSNR+coverage score vs resident reference images",
                "views_total": n_views,
                "parallelism": f"views sharded {args.views_per_gpu}/GPU, one all-gather of 16-B records",
            },
            "nominal_ray_samples_per_s": nom_all * k / elapsed,
            "rays_per_s": rays_all * k / elapsed,
            "views_scored_per_s": n_views * k / elapsed,
            "samples_evaluated_per_step_per_gpu": evaluated_per_step,
            "samples_nominal_per_step_per_gpu": nominal_per_step,
            "ranking_head": [int(x) for x in order[:8]],
            "roofline": {
                "bound": "hbm",
                "kernel": "render_queue_kernel",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": measured_traffic(args, samples_per_launch),
                "bytes_per_unit": BYTES_PER_SAMPLE,
                "units_per_launch": samples_per_launch,
                "avg_launch_ms": kernel_s * 1e3,
                "launches": launches,
                "march_avg_launch_ms": prof["march_ms"] / max(1, prof["march_launches"]),
                "mfma_tflops": samples_per_launch * 20480 / kernel_s / 1e12,
                "mfma_util_frac": samples_per_launch * 20480 / kernel_s / 2.5e15,  # of the ~2.5 PFLOP/s dense f16 peak
                "note": ("algorithmic gather bytes (512 B per evaluated sample) over the 8 TB/s HBM peak, as SURVEY 8d "
                         "prescribes; the 256^3 table (17.7 MiB) is L2 / Infinity-Cache resident, so frac > 1 is a cache "
                         "effect: `traffic` = measured fabric-side bytes per launch (FETCH_SIZE + WRITE_SIZE, separate PMC passes; "
                         "FETCH_SIZE is NOT doubled here: the guide's x2 applies to wide coalesced reads tallied as 128-B "
                         "requests, these are per-lane gathers, calibrated at one 64-B request per missing load, "
                         "profiles/r01_gather_calib.txt); the 512^3 field "
                         "(--field 512, 64 MiB table) is the HBM-bound case, frac 0.44 = the random-64-B-request ceiling "
                         "(DESIGN.md section 3)") if args.field == "256" else
                        ("64 MiB table: L2 hit rate 29 %, fabric reads ~535 B per sample; 3.5 TB/s of algorithmic bytes "
                         "against the 3.8 TB/s this GPU serves as random 64-B requests (profiles/r01_gather_calib.txt)"),
            },
            "first_hit": first_hit,
            "training": training,
        }
        if not args.no_cpu_baseline and world == 1:  # reported at N=1 only
            out["cpu_baseline"] = cpu_baseline(args, tms, scale, offset, fov_x)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
