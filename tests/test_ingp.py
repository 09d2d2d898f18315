"""instant-ngp snapshot import / export (prv_model_load_ingp / save_ingp, host twins prvh_ingp_read / write) without a
GPU.  The format is ASSUMED FROM UPSTREAM AND UNPINNED (no instant-ngp, no snapshot in the reference tree or here);
what these tests pin is that reader and writer implement the documented layout (csrc/prv_ingp.hpp) -- against an
independent writer/reader in this file (python's msgpack module + numpy, and one snapshot assembled byte by byte) --
and that everything the build cannot represent is refused with a message instead of being mis-read."""
import gzip
import struct

import numpy as np
import pytest

from nerf_prv_amd import _lib as L
from nerf_prv_amd import planner

msgpack = pytest.importorskip("msgpack")

LAYERS = [(32, 64), (64, 16), (32, 64), (64, 64), (64, 16)]  # (in, out): density 1-2, rgb 1-3


def tcnn_levels(n_levels, base, pls, log2_T):
    """tiny-cuda-nn grid.h in numpy float32: scale, resolution, entries per level"""
    out = []
    l2 = np.log2(np.float32(pls))
    for l in range(n_levels):
        scale = np.exp2(np.float32(l) * l2) * np.float32(base) - np.float32(1.0)
        res = int(np.ceil(scale)) + 1
        n = min(-(-res ** 3 // 8) * 8, 1 << log2_T)
        out.append((float(scale), res, n))
    return out


def morton(x, y, z):
    def expand(v):
        v = (v * 0x00010001) & 0xFF0000FF
        v = (v * 0x00000101) & 0x0F00F00F
        v = (v * 0x00000011) & 0xC30C30C3
        v = (v * 0x00000005) & 0x49249249
        return v
    return expand(x) | (expand(y) << 1) | (expand(z) << 2)


def make_field(rng, n_levels=8, n_features=4, log2_T=10, base=4, pls=1.45, grid=16):
    lv = tcnn_levels(n_levels, base, pls, log2_T)
    n_grid = sum(n for _, _, n in lv) * n_features
    table = rng.integers(0, 0x7BFF, n_grid, dtype=np.uint16)  # finite halves
    mlp = rng.integers(0, 0x7BFF, 10240, dtype=np.uint16)
    density = rng.random(grid ** 3).astype(np.float16) * np.float16(0.05)
    density[rng.random(grid ** 3) < 0.5] = 0
    density[rng.random(grid ** 3) < 0.05] = -1  # upstream marks cells no training view sees with a negative value
    return dict(n_levels=n_levels, n_features=n_features, log2_T=log2_T, base=base, pls=pls, grid=grid, levels=lv,
                table=table, mlp=mlp, density_xyz=density.reshape(grid, grid, grid))  # density_xyz[z][y][x]


def upstream_snapshot(f, **over):
    """the snapshot as instant-ngp lays it out (module docstring of csrc/prv_ingp.hpp), written independently"""
    params = []
    off = 0
    for n_in, n_out in LAYERS:  # canonical [in][out] -> FullyFusedMLP [out][in]
        params.append(f["mlp"][off:off + n_in * n_out].reshape(n_in, n_out).T.reshape(-1))
        off += n_in * n_out
    params.append(f["table"])
    params = np.concatenate(params).astype(np.uint16)
    g = f["grid"]
    dens = np.zeros(g ** 3, np.float16)
    for z in range(g):
        for y in range(g):
            for x in range(g):
                dens[morton(x, y, z)] = f["density_xyz"][z, y, x]
    root = {
        "encoding": {"otype": "HashGrid", "n_levels": f["n_levels"], "n_features_per_level": f["n_features"],
                     "log2_hashmap_size": f["log2_T"], "base_resolution": f["base"], "per_level_scale": float(np.float32(f["pls"]))},
        "network": {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 1},
        "rgb_network": {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 2},
        "dir_encoding": {"otype": "Composite", "nested": [{"n_dims_to_encode": 3, "otype": "SphericalHarmonics", "degree": 4},
                                                          {"otype": "Identity", "n_bins": 4, "degree": 4}]},
        "optimizer": {"otype": "Ema", "nested": {"otype": "Adam", "learning_rate": 1e-2}},  # ignored on read
        "snapshot": {"version": 1, "mode": "nerf", "n_params": int(params.size), "params_type": "__half",
                     "params_binary": params.tobytes(), "density_grid_size": g, "density_grid_binary": dens.tobytes(),
                     "nerf": {"aabb_scale": 1, "cam_pos_offset": [[0.0, 0.0, 0.0]]}, "training_step": 2500, "loss": 1e-3,
                     "camera": {"matrix": [[1.0, 0.0, 0.0, 0.0]] * 3, "fov_axis": 0}},
    }
    for path, v in over.items():
        node = root
        keys = path.split(".")
        for k in keys[:-1]:
            node = node[k]
        if v is None:
            node.pop(keys[-1], None)
        else:
            node[keys[-1]] = v
    return root


def expected_occ(f):
    d = f["density_xyz"].astype(np.float32)
    # upstream's update_density_grid_mean_and_bitfield: the mean of max(v, 0) over ALL cells -- the never-seen (-1) ones
    # add nothing to the sum but count in the denominator
    mean = np.float32(np.maximum(d, 0).astype(np.float64).sum() / d.size)
    thresh = min(0.01, float(mean))
    bits = (d > np.float32(thresh)).reshape(-1)  # z, y, x with x fastest = bit x + R*(y + R*z)
    words = np.zeros((bits.size + 31) // 32, np.uint32)
    idx = np.flatnonzero(bits)
    np.bitwise_or.at(words, idx >> 5, (np.uint32(1) << (idx & 31).astype(np.uint32)))
    return words


def test_unseen_cells_count_in_the_density_mean(tmp_path):
    """a thin field (every optical thickness below 0.01, so the threshold is the MEAN) in which most cells were never
    seen: upstream divides by all cells, so the threshold is far lower than the mean of the seen cells, and cells between
    the two are occupied"""
    rng = np.random.default_rng(11)
    f = make_field(rng, grid=16)
    d = (rng.random(16 ** 3) * 0.008).astype(np.float16)
    d[rng.random(16 ** 3) < 0.7] = -1
    f["density_xyz"] = d.reshape(16, 16, 16)
    seen = d[d >= 0].astype(np.float32)
    all_mean = float(np.maximum(d.astype(np.float32), 0).sum() / d.size)
    assert all_mean < 0.5 * float(seen.mean()) < 0.01
    between = int(((d.astype(np.float32) > all_mean) & (d.astype(np.float32) <= seen.mean())).sum())
    assert between > 100  # cells the wrong denominator would drop
    p = write(tmp_path, "thin.msgpack", upstream_snapshot(f), False)
    _, _, _, occ = planner.ingp_read(p)
    assert np.array_equal(occ, expected_occ(f))
    assert int(np.unpackbits(occ.view(np.uint8)).sum()) == int((d.astype(np.float32) > np.float32(all_mean)).sum())


def test_hostile_snapshots_are_refused_before_they_eat_the_host(tmp_path):
    """a few MB that promise GBs: a msgpack array of a million-and-one one-byte values (each would become a ~150-byte
    tree node); a gzip bomb is cut off at the inflated-size cap"""
    n = (1 << 20) + 1
    raw = b"\xdd" + n.to_bytes(4, "big") + b"\x00" * n
    p = tmp_path / "items.msgpack"
    p.write_bytes(raw)
    with pytest.raises(Exception, match="values"):
        planner.ingp_read(p)


def write(tmp_path, name, root, zipped):
    raw = msgpack.packb(root, use_bin_type=True)
    p = tmp_path / name
    p.write_bytes(gzip.compress(raw) if zipped else raw)
    return p


@pytest.mark.parametrize("shape", [dict(), dict(n_levels=16, n_features=2, log2_T=9, base=3, pls=1.3819), dict(log2_T=19, base=16, pls=1.38191, grid=8)])
@pytest.mark.parametrize("zipped", [True, False])
def test_reader_maps_the_upstream_layout_to_the_canonical_arrays(tmp_path, shape, zipped):
    f = make_field(np.random.default_rng(7), **shape)
    p = write(tmp_path, "snap.ingp" if zipped else "snap.msgpack", upstream_snapshot(f), zipped)
    d, table, mlp, occ = planner.ingp_read(p)
    assert (d.n_levels, d.n_features, d.log2_hashmap, d.base_res, d.occ_res) == (f["n_levels"], f["n_features"], f["log2_T"], f["base"], f["grid"])
    assert d.per_level_scale == np.float32(f["pls"]) and d.density_bias == 0.0
    assert np.array_equal(table, f["table"])  # the grid is the same layout on both sides
    assert np.array_equal(mlp, f["mlp"])      # [out][in] -> [in][out], layer by layer
    assert np.array_equal(occ, expected_occ(f))
    assert 0 < int(np.unpackbits(occ.view(np.uint8)).sum()) < f["grid"] ** 3
    # the level geometry the library derives from that descriptor is tiny-cuda-nn's (float32 recipe), level by level
    from oracle import oracle as orc

    lv, total = orc.levels(orc.desc(n_levels=d.n_levels, n_features=d.n_features, log2_hashmap=d.log2_hashmap, base_res=d.base_res,
                                    finest_res=d.finest_res, occ_res=d.occ_res, per_level_scale=d.per_level_scale))
    want = f["levels"]
    assert total == sum(n for _, _, n in want)
    for l, (scale, res, n) in enumerate(want):
        assert (lv[l].res, lv[l].size) == (res, n) and lv[l].hashed == int(res ** 3 > (1 << f["log2_T"]))
        # exp2f / log2f of two math libraries differ in the last place, amplified by the level index (as the host's and
        # CUDA's do upstream): a few ulp, i.e. a sample moves by < 1e-6 of a cell
        assert abs(lv[l].scale - np.float32(scale)) <= 1e-6 * scale


def test_writer_emits_the_upstream_layout_and_round_trips(tmp_path):
    f = make_field(np.random.default_rng(11), log2_T=12, base=8, pls=1.5)
    src = write(tmp_path, "in.msgpack", upstream_snapshot(f), False)
    d, table, mlp, occ = planner.ingp_read(src)
    for name in ("out.ingp", "out.msgpack"):
        out = tmp_path / name
        planner.ingp_write(out, d, table, mlp, occ)
        raw = out.read_bytes()
        if name.endswith(".ingp"):
            assert raw[:2] == b"\x1f\x8b"
            raw = gzip.decompress(raw)
        root = msgpack.unpackb(raw, raw=False)
        enc, snap = root["encoding"], root["snapshot"]
        assert (enc["otype"], enc["n_levels"], enc["n_features_per_level"], enc["log2_hashmap_size"], enc["base_resolution"]) == \
            ("HashGrid", 8, 4, 12, 8) and enc["per_level_scale"] == 1.5
        assert root["network"]["n_hidden_layers"] == 1 and root["rgb_network"]["n_hidden_layers"] == 2
        assert snap["mode"] == "nerf" and snap["params_type"] == "__half" and snap["nerf"]["aabb_scale"] == 1 and snap["density_grid_size"] == 16
        params = np.frombuffer(snap["params_binary"], np.uint16)
        assert snap["n_params"] == params.size == 10240 + table.size
        off = 0
        for n_in, n_out in LAYERS:  # [out][in] in the file
            assert np.array_equal(params[off:off + n_in * n_out].reshape(n_out, n_in), mlp[off:off + n_in * n_out].reshape(n_in, n_out).T)
            off += n_in * n_out
        assert np.array_equal(params[10240:], table)
        dens = np.frombuffer(snap["density_grid_binary"], np.float16)
        g = 16
        for (x, y, z) in ((0, 0, 0), (3, 5, 7), (15, 15, 15), (1, 0, 0), (0, 1, 0), (0, 0, 1)):
            bit = x + g * (y + g * z)
            assert (dens[morton(x, y, z)] > 0) == bool((occ[bit >> 5] >> (bit & 31)) & 1)
        d2, t2, m2, o2 = planner.ingp_read(out)  # and back
        assert bytes(d2) == bytes(d) and np.array_equal(t2, table) and np.array_equal(m2, mlp) and np.array_equal(o2, occ)


def test_a_snapshot_assembled_byte_by_byte(tmp_path):
    """no msgpack library on the writing side: fixmap / fixstr / positive fixint / uint16 / float32 / bin32 / fixarray by hand"""
    f = make_field(np.random.default_rng(3), log2_T=4, base=2, pls=1.25, grid=8)
    ref = upstream_snapshot(f)
    params, dens = ref["snapshot"]["params_binary"], ref["snapshot"]["density_grid_binary"]

    def s(t):
        b = t.encode()
        assert len(b) < 32
        return bytes([0xA0 | len(b)]) + b

    def fixmap(items):
        assert len(items) < 16
        return bytes([0x80 | len(items)]) + b"".join(s(k) + v for k, v in items)

    def u(v):
        return bytes([v]) if v < 128 else b"\xcd" + struct.pack(">H", v)

    def bin32(b):
        return b"\xc6" + struct.pack(">I", len(b)) + b

    mlp1 = fixmap([("otype", s("FullyFusedMLP")), ("activation", s("ReLU")), ("output_activation", s("None")), ("n_neurons", u(64)), ("n_hidden_layers", u(1))])
    mlp2 = fixmap([("otype", s("FullyFusedMLP")), ("activation", s("ReLU")), ("output_activation", s("None")), ("n_neurons", u(64)), ("n_hidden_layers", u(2))])
    enc = fixmap([("otype", s("HashGrid")), ("n_levels", u(8)), ("n_features_per_level", u(4)), ("log2_hashmap_size", u(4)), ("base_resolution", u(2)),
                  ("per_level_scale", b"\xca" + struct.pack(">f", 1.25))])
    dire = fixmap([("otype", s("SphericalHarmonics")), ("degree", u(4))])
    snap = fixmap([("version", u(1)), ("mode", s("nerf")), ("n_params", u(len(params) // 2)), ("params_type", s("__half")), ("params_binary", bin32(params)),
                   ("density_grid_size", u(8)), ("density_grid_binary", bin32(dens)), ("nerf", fixmap([("aabb_scale", u(1))])),
                   ("aabb", bytes([0x92]) + b"\xca" + struct.pack(">f", 0.0) + b"\xca" + struct.pack(">f", 1.0))])
    raw = fixmap([("encoding", enc), ("network", mlp1), ("rgb_network", mlp2), ("dir_encoding", dire), ("snapshot", snap)])
    p = tmp_path / "hand.msgpack"
    p.write_bytes(raw)
    d, table, mlp, occ = planner.ingp_read(p)
    assert (d.n_levels, d.n_features, d.log2_hashmap, d.base_res, d.occ_res, d.per_level_scale) == (8, 4, 4, 2, 8, 1.25)
    assert np.array_equal(table, f["table"]) and np.array_equal(mlp, f["mlp"]) and np.array_equal(occ, expected_occ(f))


@pytest.mark.parametrize("change,exc,needle", [
    ({"snapshot.nerf.aabb_scale": 4}, ValueError, "aabb_scale 4"),
    ({"encoding.n_levels": 8, "encoding.n_features_per_level": 2}, ValueError, "!= 32"),
    ({"network.n_neurons": 128}, ValueError, "n_neurons 64"),
    ({"rgb_network.n_hidden_layers": 3}, ValueError, "hidden"),
    ({"network.activation": "Sigmoid"}, ValueError, "ReLU"),
    ({"network.otype": "MegaMLP"}, ValueError, "otype"),
    ({"encoding.otype": "Frequency"}, ValueError, "hash grid"),
    ({"encoding.type": "Dense"}, ValueError, "grid type"),
    ({"dir_encoding": {"otype": "SphericalHarmonics", "degree": 3}}, ValueError, "degree 4"),
    ({"snapshot.mode": "sdf"}, ValueError, "not nerf"),
    ({"snapshot.params_type": "float"}, ValueError, "__half"),
    ({"snapshot.params_binary": b"\0" * 100}, IOError, "params_binary holds"),
    ({"snapshot.n_params": 17}, IOError, "n_params 17"),
    ({"snapshot.density_grid_binary": b"\0" * 10}, IOError, "density_grid_binary"),
    ({"snapshot": None}, IOError, "missing"),
    ({"encoding.per_level_scale": 40.0}, ValueError, "exceeds 4096"),
])
def test_what_cannot_be_represented_is_refused_with_a_message(tmp_path, change, exc, needle):
    f = make_field(np.random.default_rng(5))
    p = write(tmp_path, "bad.msgpack", upstream_snapshot(f, **change), False)
    with pytest.raises(exc, match=needle):
        planner.ingp_read(p)


def test_malformed_files_are_errors_not_crashes(tmp_path):
    f = make_field(np.random.default_rng(9))
    good = msgpack.packb(upstream_snapshot(f), use_bin_type=True)
    rng = np.random.default_rng(1)
    cases = {"empty": b"", "not_msgpack": b"\xc1\xc1\xc1", "array_on_top": msgpack.packb([1, 2, 3]), "truncated": good[: len(good) // 2],
             "bad_gzip": b"\x1f\x8b" + bytes(rng.integers(0, 256, 200, dtype=np.uint8)), "gzip_truncated": gzip.compress(good)[:-40],
             "huge_array_header": b"\xdd\xff\xff\xff\xff", "huge_map_header": b"\xdf\x7f\xff\xff\xff\xa1a", "deep": b"\x91" * 5000 + b"\xc0"}
    for name, data in cases.items():
        p = tmp_path / f"{name}.ingp"
        p.write_bytes(data)
        with pytest.raises(IOError):
            planner.ingp_read(p)
    for k in range(300):  # random single-byte damage never crashes the reader
        b = bytearray(good)
        b[int(rng.integers(0, min(len(b), 4000)))] = int(rng.integers(0, 256))
        p = tmp_path / "fuzz.msgpack"
        p.write_bytes(bytes(b))
        try:
            planner.ingp_read(p)
        except (IOError, ValueError):
            pass
    with pytest.raises(IOError):
        planner.ingp_read(tmp_path / "does_not_exist.ingp")


def test_fields_without_a_per_level_scale_are_not_written_as_snapshots(tmp_path):
    """a base_res/finest_res field has level scales no per_level_scale reproduces: writing it would silently move every
    sample inside its cells on the reading side"""
    d = L.FieldDesc(n_levels=8, n_features=4, log2_hashmap=10, base_res=4, finest_res=40, occ_res=16, density_bias=0.0, table_amp=0.0,
                    per_level_scale=0.0)
    from oracle import oracle as orc

    lv, total = orc.levels(orc.desc(n_levels=8, n_features=4, log2_hashmap=10, base_res=4, finest_res=40, occ_res=16))
    with pytest.raises(ValueError, match="per_level_scale"):
        planner.ingp_write(tmp_path / "x.ingp", d, np.zeros(total * 4, np.uint16), np.zeros(10240, np.uint16), np.zeros(128, np.uint32))
