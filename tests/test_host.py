"""Host-side logic that needs no GPU: the C ABI library loads and exports every entry point of
include/caster_hip.h, compute entry points fail loudly without a device, the C host's scenes,
camera, flattening and BMP writer behave as the reference's host code, and the product never
touches the oracle."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_entry_points():
    header = open(os.path.join(ROOT, "include", "caster_hip.h")).read()
    return sorted(set(re.findall(r"\b(caster_hip_[a-z_0-9]+)\s*\(", header)))


def test_c_abi_exports_every_declared_symbol(ca):
    lib = ca.load_caster_lib()
    names = _declared_entry_points()
    assert len(names) >= 20
    for name in names:
        assert getattr(lib, name) is not None, name
    out = subprocess.check_output(["nm", "-D", "--defined-only", ca.lib_paths()["caster"]], text=True)
    exported = set(re.findall(r" T (caster_hip_[a-z_0-9]+)", out))
    assert set(names) <= exported
    # plain C ABI: nothing but caster_hip_* is exported with C linkage from the API unit
    assert not [s for s in exported if not s.startswith("caster_hip_")]


def test_no_device_is_a_loud_error(ca):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the no-device path cannot be exercised")
    with pytest.raises(ca.CasterHipError) as e:
        ca.CasterHip(0)
    assert "no CPU fallback" in str(e.value) or "HIP" in str(e.value)
    lib = ca.load_caster_lib()
    n = ctypes.c_int(-1)
    rc = lib.caster_hip_device_count(ctypes.byref(n))
    assert rc != 0 or n.value == 0
    assert lib.caster_hip_render(None, None, 1, 1, 0, 1, 1, 0, 0, None) != 0       # NULL ctx: argument error
    assert b"NULL" in lib.caster_hip_last_error()


def test_product_never_references_the_oracle():
    bad = []
    for base in ("craytracer_amd", "include", "tools"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".c", ".h", ".hpp", ".hip", ".cpp", "Makefile")):
                    text = open(os.path.join(dirpath, f), errors="replace").read()
                    if re.search(r"oracle|orc_", text):
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_scene_0_matches_reference_setup(ca):
    """src/main.c:14-55: ground checker sphere, glass sphere, third sphere, emissive xy rect, in that order."""
    fs = ca.scene_0()
    assert fs.objects["type"].tolist() == [0, 0, 0, 1]
    assert np.allclose(fs.objects["g"][0, :4], [0, -1000, 0, 1000])
    assert np.allclose(fs.objects["g"][1, :4], [4, 1, 0, 1])
    assert np.allclose(fs.objects["g"][2, :4], [4, 1, 3, 1])
    assert np.allclose(fs.objects["g"][3, :5], [4, 2, -2, 2, 2])
    m = fs.materials
    assert m["type"].tolist() == [0, 2, 0, 3] and m["tex_type"].tolist() == [1, 0, 0, 0]
    assert np.allclose(m["color0"][0], 1) and np.allclose(m["color1"][0], 0)
    assert m["n_refract"][1] == np.float32(1.5)
    assert np.allclose(m["color0"][3], 2)
    tex = ca.scene_0(textured=True)
    assert tex.materials["tex_type"][2] == 2 and tex.materials["image"][2] == 0 and tex.images[0].shape == (128, 256, 3)


def test_rtiow_and_stress_scenes(ca):
    a, b = ca.scene_rtiow(1), ca.scene_rtiow(1)
    assert np.array_equal(a.objects, b.objects) and np.array_equal(a.materials, b.materials)      # deterministic
    c = ca.scene_rtiow(2)
    assert not np.array_equal(a.objects["g"], c.objects["g"][: len(a.objects)]) if len(a.objects) == len(c.objects) else True
    assert 470 <= len(a.objects) <= 489 and (a.objects["type"] == 0).all()
    assert np.allclose(a.objects["g"][0, :4], [0, -1000, 0, 1000])
    assert np.allclose(a.objects["g"][-1, :4], [0, 0, 0, 100000]) and a.materials["type"][a.objects["material"][-1]] == 3
    # the three big spheres: glass, diffuse (0.4,0.2,0.1), metal (0.7,0.6,0.5) fuzz 0 -- src/main.c:24-26
    big = a.objects[-4:-1]
    assert np.allclose(big["g"][:, 3], 1.0)
    mats = a.materials[big["material"]]
    assert mats["type"].tolist() == [2, 0, 1]
    assert np.allclose(mats["color0"][1], [0.4, 0.2, 0.1]) and np.allclose(mats["color0"][2], [0.7, 0.6, 0.5])
    small = a.objects[1:-4]
    assert np.allclose(small["g"][:, 3], 0.2) and np.allclose(small["g"][:, 1], 0.2)
    kinds = np.bincount(a.materials["type"][small["material"]], minlength=3) / len(small)
    assert 0.7 < kinds[0] < 0.9 and 0.08 < kinds[1] < 0.22 and 0.01 < kinds[2] < 0.1
    s = ca.scene_stress(1000, 2)
    assert len(s.objects) == 1002 and np.allclose(s.objects["g"][-1, 3], 100000)
    r = s.objects["g"][1:-1, 3]
    assert r.min() >= 0.1 and r.max() <= 0.5


def test_perlin_scene_and_noise_tables(ca):
    n1, n2, n3 = ca.make_noise(1), ca.make_noise(1), ca.make_noise(2)
    assert n1.tobytes() == n2.tobytes() and n1.tobytes() != n3.tobytes()
    assert np.allclose(np.linalg.norm(n1["ranvec"], axis=1), 1.0, atol=1e-6)
    for t in ("perm_x", "perm_y", "perm_z"):
        assert sorted(n1[t].tolist()) == list(range(256))                 # a permutation of 0..255
    assert n1["perm_x"].tolist() != n1["perm_y"].tolist()
    fs = ca.scene_perlin(1)
    assert fs.noise is not None and fs.noise.tobytes() == ca.make_noise(1).tobytes()
    rt = ca.scene_rtiow(1)
    assert abs(len(fs.objects) - len(rt.objects)) < 8 and np.array_equal(fs.objects["g"][-4:], rt.objects["g"][-4:])
    assert np.array_equal(fs.objects["g"][0], rt.objects["g"][0])
    tex = fs.materials["tex_type"][fs.objects["material"]]
    assert tex[0] == 4 and (tex == 4).sum() > 300 and tex[-1] == 0
    noisy = fs.materials[fs.materials["tex_type"] == 4]
    assert (noisy["type"] == 0).all() and noisy["color1"][:, 0].min() >= 4.0 and noisy["color1"][:, 0].max() <= 40.0
    assert ca.scene_rtiow(1).noise is None


def test_host_camera_equals_oracle_camera(ca, orc):
    for aspect in (16 / 9, 4 / 3, 1.0):
        assert np.array_equal(ca.default_camera(aspect), orc.default_camera(aspect))
    a = ca.calculate_cam_info((1, 2, 3), (0, 0.5, -1), (0, 1, 0), 40.0, 1.5, 0.3, 4.0)
    b = orc.calculate_cam_info((1, 2, 3), (0, 0.5, -1), (0, 1, 0), 40.0, 1.5, 0.3, 4.0)
    assert np.array_equal(a, b)


def test_host_bmp_writer_bytes(ca, orc):
    rng = np.random.default_rng(0)
    for w, h in ((3, 2), (5, 4), (16, 9), (7, 1)):
        rgba = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        assert ca.bmp_encode(rgba) == orc.bmp_bytes(rgba)


def test_cli_rejects_bad_arguments(ca):
    exe = os.path.join(os.path.dirname(ca.lib_paths()["host"]), "craytracer")
    assert os.path.exists(exe)
    p = subprocess.run([exe], capture_output=True, text=True)
    assert p.returncode == 2 and "usage" in p.stderr
    p = subprocess.run([exe, "90", "1", "5", "1", "--scene", "nope"], capture_output=True, text=True)
    assert p.returncode != 0


def test_sample_partition():
    from craytracer_amd.frame import sample_partition

    for spp in (0, 1, 7, 500, 2000):
        for world in (1, 2, 3, 4, 8):
            parts = [sample_partition(spp, world, r) for r in range(world)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == spp
            for (b0, c0), (b1, _) in zip(parts, parts[1:]):
                assert b0 + c0 == b1
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1
    with pytest.raises(ValueError):
        sample_partition(10, 2, 2)


def test_sample_range_lists_and_checkpoint_file(tmp_path, ca):
    """The bookkeeping of progressive rendering: which samples a frame holds, which are left, how the
    rest is dealt out to the jobs, and the checkpoint file's round trip and refusals."""
    import craytracer_amd.binding as b

    host = b.load_host_lib()
    R = np.dtype([("begin", "<i4"), ("count", "<i4")])

    def arr(pairs, cap=None):
        a = np.zeros(cap or max(len(pairs), 1), R)
        for k, (s, c) in enumerate(pairs):
            a[k] = (s, c)
        return a

    def ptr(a):
        return a.ctypes.data_as(ctypes.c_void_p)

    def as_set(a, n):
        s = set()
        for k in range(n):
            s.update(range(int(a["begin"][k]), int(a["begin"][k] + a["count"][k])))
        return s

    rng = np.random.default_rng(5)
    for _ in range(200):
        pairs = [(int(rng.integers(0, 60)), int(rng.integers(0, 9))) for _ in range(int(rng.integers(0, 7)))]
        a = arr(pairs)
        want = as_set(a, len(pairs))
        n = host.crt_ranges_normalize(ptr(a), len(pairs))
        assert as_set(a, n) == want and host.crt_ranges_total(ptr(a), n) == len(want)
        assert all(a["begin"][k] + a["count"][k] < a["begin"][k + 1] for k in range(n - 1))       # sorted, merged
        spp = int(rng.integers(0, 80))
        rest = arr([], 16)
        m = host.crt_ranges_complement(ptr(a), n, spp, ptr(rest), 16)
        assert as_set(rest, m) == set(range(spp)) - want
        # dealing the rest out to G jobs: disjoint slices, in order, sizes within one of each other
        total = host.crt_ranges_total(ptr(rest), m)
        G = int(rng.integers(1, 5))
        got = []
        for g in range(G):
            first, last = total * g // G, total * (g + 1) // G
            part = arr([], 16)
            k = host.crt_ranges_slice(ptr(rest), m, first, last - first, ptr(part), 16)
            s = as_set(part, k)
            assert len(s) == last - first == host.crt_ranges_total(ptr(part), k)
            got.append(s)
        assert set().union(*got) == set(range(spp)) - want and sum(len(s) for s in got) == total
    assert host.crt_ranges_complement(ptr(arr([(2, 1), (5, 1), (8, 1)])), 3, 20, ptr(arr([], 2)), 2) == -1      # cap too small

    # the file: header fields, ranges and sums survive; truncated, padded and foreign files are refused
    class CP(ctypes.Structure):
        _fields_ = [("width", ctypes.c_uint32), ("height", ctypes.c_uint32), ("seed", ctypes.c_uint32),
                    ("strict", ctypes.c_int32), ("max_depth", ctypes.c_int32), ("scene_hash", ctypes.c_uint64),
                    ("n_ranges", ctypes.c_int32), ("ranges", ctypes.c_int32 * 512), ("sums", ctypes.c_void_p)]

    host.crt_checkpoint_read.restype = ctypes.POINTER(CP)
    sums = rng.random((3, 5, 4), dtype=np.float32)
    cp = CP(5, 3, 77, 1, 50, 0x1234567890ABCDEF, 2)
    cp.ranges[0:4] = [10, 4, 0, 6]
    cp.sums = sums.ctypes.data
    path = str(tmp_path / "a.ck").encode()
    assert host.crt_checkpoint_write(path, ctypes.byref(cp)) == 0
    back = host.crt_checkpoint_read(path)
    assert back and (back.contents.width, back.contents.height, back.contents.seed, back.contents.strict,
                     back.contents.max_depth, back.contents.scene_hash) == (5, 3, 77, 1, 50, 0x1234567890ABCDEF)
    assert back.contents.n_ranges == 2 and list(back.contents.ranges[0:4]) == [0, 6, 10, 4]     # normalized on reading
    got = np.ctypeslib.as_array(ctypes.cast(back.contents.sums, ctypes.POINTER(ctypes.c_float)), shape=(3, 5, 4))
    assert np.array_equal(got, sums)
    host.crt_checkpoint_free(back)
    data = open(path, "rb").read()
    for name, blob in (("short", data[:-4]), ("long", data + b"\0"), ("magic", b"X" + data[1:]), ("empty", b"")):
        p = str(tmp_path / name).encode()
        open(p, "wb").write(blob)
        assert not host.crt_checkpoint_read(p)
    assert not host.crt_checkpoint_read(str(tmp_path / "missing").encode())


class _SumStep(ctypes.Structure):
    _fields_ = [("dst", ctypes.c_int), ("src", ctypes.c_int), ("round", ctypes.c_int)]


def _fence_sum(host, device):
    """Runs the fence's summation plan for a job -> device map on integer stand-ins for the buffers
    (job g holds 1 << g, so any sum names exactly the jobs it contains)."""
    n = len(device)
    dev = (ctypes.c_int * n)(*device)
    lead = (ctypes.c_int * 64)()
    steps = (_SumStep * 64)()
    n_steps = ctypes.c_int()
    n_lead = host.crt_sum_plan_local(dev, n, lead, steps, ctypes.byref(n_steps))
    buf = [1 << g for g in range(n)]
    for k in range(n_steps.value):
        assert steps[k].round == 0 and device[steps[k].dst] == device[steps[k].src]      # an add kernel never crosses devices
        buf[steps[k].dst] += buf[steps[k].src]
    leads = [lead[l] for l in range(n_lead)]
    return buf, leads, steps


def test_fence_sum_plan_leads_and_tree_order(ca):
    """The order in which the fence adds the jobs' buffers, checked as data: the first job of each
    device leads it, every job's buffer is added exactly once, job 0 ends up with the frame; the
    pairwise tree (the route without RCCL) adds disjoint pairs per round and finishes in ceil(log2 G) rounds."""
    host = ca.load_host_lib()
    for device in ([0], [0, 0, 0], [0, 1, 2, 3], [0, 1, 2, 3, 4, 5, 6, 7], [0, 1, 0, 1, 0], [3, 3, 1, 3, 1, 0, 2, 2, 0],
                   list(range(8)) * 2, [g % 3 for g in range(11)]):
        n = len(device)
        buf, leads, steps = _fence_sum(host, device)
        assert leads[0] == 0
        assert [device[l] for l in leads] == list(dict.fromkeys(device))        # one lead per device, in order of first use
        assert sorted(set(device)) == sorted(device[l] for l in leads)
        for d in set(device):                                                    # a lead holds exactly its device's jobs
            l = [x for x in leads if device[x] == d][0]
            assert buf[l] == sum(1 << g for g in range(n) if device[g] == d)
        # the tree over the leads
        n_lead = len(leads)
        n_tree = host.crt_sum_plan_tree(n_lead, steps)
        assert n_tree == n_lead - 1
        val = [buf[l] for l in leads]
        rounds = {}
        for k in range(n_tree):
            rounds.setdefault(steps[k].round, []).append((steps[k].dst, steps[k].src))
        assert sorted(rounds) == list(range(1, len(rounds) + 1))
        assert len(rounds) == (0 if n_lead == 1 else int(np.ceil(np.log2(n_lead))))
        for r in sorted(rounds):
            touched = [x for pair in rounds[r] for x in pair]
            assert len(touched) == len(set(touched))                             # the adds of a round are independent
            for dst, src in rounds[r]:
                val[dst] += val[src]
                val[src] = 0                                                     # consumed: must not be added again
        assert val[0] == (1 << n) - 1                                            # every job exactly once, in job 0's buffer
