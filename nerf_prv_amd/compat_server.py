"""Flag-file compatibility server: lets the UNMODIFIED reference executable drive this build.

Protocol (reference side): `NBV_Net_Labeler::train_by_instantNGP` writes
`<instant_ngp_path>/interact/run_with_c++.py` containing one `os.system('python .../run.py ...')`
line, touches `interact/ready_c++.txt`, then polls `interact/ready_py.txt` once a second
(main.cpp:1661-1701).  The reference's `train_server.py:7-14` polls the other way, deletes the
flag, runs the script and touches `ready_py.txt`.

This server honours the same files but never spawns anything: it parses the run.py command line
out of the script and answers it in-process through the C ABI:
  --screenshot_transforms J --screenshot_dir D   -> render every frame of J, write D/<basename>.png
                                                    (run.py:284-309)
  --test_transforms J --save_metrics M           -> evaluate against reference images, write M
                                                    (run.py:213-277)
  --train --n_steps N --scene J                  -> a fresh field trained N steps in process on J's views
                                                    (run.py:109, 185-208; `train_desc=` names the field),
                                                    unless a `load_model(scene, ctx) -> slot` callable supplies
                                                    weights from elsewhere (e.g. `.prvf` files)
PNG encoding uses PIL (present in the image); the byte rule is the library's (prv_render_rgba8).
"""
import json
import os
import re
import shlex
import time

import numpy as np

from . import api

CMD_RE = re.compile(r"os\.system\('(.*)'\)")


def parse_command(script_text):
    """the run.py arguments of the one-line script the reference writes (main.cpp:1665-1688)"""
    m = CMD_RE.search(script_text)
    if not m:
        raise ValueError("no os.system('...') line in run_with_c++.py")
    toks = shlex.split(m.group(1))
    args, i = {"flags": set()}, 0
    while i < len(toks):
        t = toks[i]
        if t.startswith("--"):
            if i + 1 < len(toks) and not toks[i + 1].startswith("--"):
                args[t[2:]] = toks[i + 1]
                i += 2
                continue
            args["flags"].add(t[2:])
        i += 1
    return args


def load_reference_images(ctx, test_json):
    """the reference images of a transforms.json, as instant-ngp's load_training_data takes them: each frame's
    `file_path` PNG relative to the json (run.py:238), sRGB -> linear, premultiplied alpha, float32 on the
    device.  ASSUMED from upstream (the loader is inside pyngp): standard sRGB EOTF, straight-alpha PNGs."""
    from PIL import Image

    with open(test_json) as f:
        meta = json.load(f)
    base = os.path.dirname(os.path.abspath(test_json))
    imgs = []
    for frame in meta["frames"]:
        path = os.path.join(base, frame["file_path"])
        if not os.path.splitext(path)[1]:
            path += ".png"
        a = np.asarray(Image.open(path).convert("RGBA"), np.float32) / 255.0
        rgb = np.where(a[..., :3] <= 0.04045, a[..., :3] / 12.92, ((a[..., :3] + 0.055) / 1.055) ** 2.4)
        imgs.append(np.concatenate([rgb * a[..., 3:4], a[..., 3:4]], axis=-1).astype(np.float32))
    return ctx.torch.from_numpy(np.stack(imgs)).to(ctx.device)


def load_dataset_bytes(ctx, scene_json):
    """the images of a dataset json as the trainer takes them: [n, h, w, 4] straight-alpha sRGB bytes on the
    device (what testbed.load_training_data uploads, run.py:109)"""
    from PIL import Image

    with open(scene_json) as f:
        meta = json.load(f)
    base = os.path.dirname(os.path.abspath(scene_json))
    imgs = []
    for frame in meta["frames"]:
        path = os.path.join(base, frame["file_path"])
        if not os.path.splitext(path)[1]:
            path += ".png"
        imgs.append(np.asarray(Image.open(path).convert("RGBA"), np.uint8))
    return ctx.torch.from_numpy(np.stack(imgs)).to(ctx.device)


def train_scene(ctx, slot, scene_json, n_steps, desc, seed=0x1234, opts=None):
    """`--train --n_steps N --scene J` in process (run.py:109, 185-208): fresh field, N optimiser steps on the
    dataset's own cameras and images; returns the per-step losses"""
    ctx.fresh_model(slot, desc, seed)
    cams = ctx.cameras_from_dataset_json(scene_json)
    tr = api.Trainer(ctx, slot, cams, load_dataset_bytes(ctx, scene_json), opts)
    losses = tr.steps(int(n_steps))
    tr.close()
    cams.close()
    return losses


class CompatServer:
    def __init__(self, interact_dir, ctx, load_model=None, samples_per_ray=0, screenshot_spp=16, reference_images=None,
                 train_desc=None, train_opts=None, train_seed=0x1234):
        self.dir, self.ctx = interact_dir, ctx
        # load_model(scene_json, ctx) -> slot supplies weights from elsewhere; without it the request's
        # `--train --n_steps N --scene J` is carried out in process on slot 0
        self.load_model = load_model or self._train
        self.train_desc, self.train_opts, self.train_seed = train_desc, train_opts, train_seed
        self.last_losses = None
        self.samples_per_ray, self.spp = samples_per_ray, screenshot_spp  # run.py:48 default spp 16
        # callable(test_json) -> device tensor, for --test_transforms; default: the PNGs the json points at
        self.reference_images = reference_images or (lambda test_json: load_reference_images(ctx, test_json))

    def _train(self, scene, ctx):
        if self.train_desc is None:
            raise api.PrvError(api.L.PRV_E_STATE, "no load_model callback and no train_desc: nothing to render with")
        n_steps = int(self._args.get("n_steps", 35000))  # run.py:177-178 default when --n_steps is absent
        self.last_losses = train_scene(ctx, 0, scene, n_steps, self.train_desc, self.train_seed, self.train_opts)
        return 0

    def serve_one(self, args):
        self._args = args
        slot = self.load_model(args.get("scene"), self.ctx)  # load_training_data + training, or supplied weights
        if "screenshot_transforms" in args:
            from PIL import Image

            with open(args["screenshot_transforms"]) as f:
                ref_transforms = json.load(f)
            cams = self.ctx.cameras_from_json(args["screenshot_transforms"])
            w, h = int(ref_transforms["w"]), int(ref_transforms["h"])  # run.py:304
            opts = api.engine_render_opts(w, h, self.samples_per_ray, self.spp, 0.01, background=(0.0, 0.0, 0.0, 1.0))
            u8, _ = self.ctx.render_rgba8(slot, cams, None, opts)
            u8 = u8.cpu().numpy()
            os.makedirs(args["screenshot_dir"], exist_ok=True)
            for frame, img in zip(ref_transforms["frames"], u8):
                name = os.path.basename(frame["file_path"])  # run.py:297
                if not os.path.splitext(name)[1]:
                    name += ".png"
                Image.fromarray(img, "RGBA").save(os.path.join(args["screenshot_dir"], name))
            cams.close()
        elif "test_transforms" in args:
            from . import planner

            # the dataset's own cameras: fl_x/fl_y, principal point, lens (run.py:238-242)
            cams = self.ctx.cameras_from_dataset_json(args["test_transforms"])
            w, h = cams.size
            gt = self.reference_images(args["test_transforms"])
            opts = api.engine_render_opts(w, h, self.samples_per_ray, 1, 1e-4, background=(0.0, 0.0, 0.0, 1.0))  # run.py:226-235
            psnr, ssim = self.ctx.evaluate(slot, cams, None, opts, gt)
            planner.write_metrics(args["save_metrics"], psnr, ssim)
            cams.close()

    def poll_once(self):
        """one iteration of the train_server.py loop; returns True when a request was served"""
        flag = os.path.join(self.dir, "ready_c++.txt")
        if not os.path.exists(flag):
            return False
        os.remove(flag)  # train_server.py:11
        with open(os.path.join(self.dir, "run_with_c++.py")) as f:
            args = parse_command(f.read())
        self.serve_one(args)
        open(os.path.join(self.dir, "ready_py.txt"), "w").close()  # train_server.py:13
        return True

    def serve_forever(self, poll_seconds=0.1):
        while True:
            if not self.poll_once():
                time.sleep(poll_seconds)  # train_server.py:9
