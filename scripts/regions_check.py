"""dev: the reference round's records (and member images' checksums) under PRV_SPATIAL_REGIONS=0 / 1 must be identical"""
import hashlib, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    from nerf_prv_amd import api, planner
    n_views, members = int(sys.argv[2]), 5
    ctx = api.Context(0)
    fd = api.L.FieldDesc(**api.FIELD_256)
    for e in range(members):
        ctx.synthetic_model(2 + e, fd, 0x5EED0001 + 16 + e)
    pts = planner.hemisphere_generate(n_views)
    fov = 2.0 * np.arctan(0.5 * 1280 / 915.606689453125)
    tms, scale, offset = planner.hemisphere_transforms(pts, 0.3, 0.1, [1e-10] * 3)
    cams = ctx.cameras_from_matrices(tms, fov, 80, 45, scale, offset)
    opts = api.engine_render_opts(80, 45, 0, 16, 0.01, background=(0, 0, 0, 1))
    rec, st = ctx.score_views(api.L.SCORE_ENSEMBLE_RGB_DENSITY, list(range(2, 2 + members)), cams, None, opts, want_stats=True)
    u8, _ = ctx.render_rgba8(2, cams, None, opts)
    print("RESULT", hashlib.sha256(rec.tobytes()).hexdigest()[:16], hashlib.sha256(u8.cpu().numpy().tobytes()).hexdigest()[:16], int(st.samples_evaluated), int(st.samples_live),
          int(ctx.argmax(rec, np.arange(n_views))))
else:
    for n in (20, 100, 540):
        outs = []
        for v in ("0", "1"):
            r = subprocess.run([sys.executable, __file__, "child", str(n)], env=dict(os.environ, PRV_SPATIAL_REGIONS=v), capture_output=True, text=True)
            outs.append([l for l in r.stdout.splitlines() if l.startswith("RESULT")][-1] if "RESULT" in r.stdout else r.stderr[-300:])
        print(n, "views:", "SAME" if outs[0] == outs[1] else "DIFFERENT", outs)
