"""Host layer (libpthost) and the C-ABI surface, on the CPU: exported symbols, world building and flattening
against the oracle's tree, the reference's error behaviour, the OBJ reader, the sharding rules."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import scenes_util as su
from rust_pathtracer_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(", text)))


def test_libraries_export_every_declared_symbol(pt):
    gpu = _declared("ptgpu.h", "pt_")
    host = _declared("pthost.h", "pth_")
    assert len(gpu) >= 15 and len(host) >= 15
    for name in gpu:
        assert hasattr(pt.libptgpu, name), f"libptgpu.so does not export {name}"
    for name in host:
        assert hasattr(pt.libpthost, name), f"libpthost.so does not export {name}"
    # and the Python binding prototypes cover exactly the declared set
    assert sorted(api.PTGPU_SYMBOLS) == gpu
    assert sorted(api.PTHOST_SYMBOLS) == host


def test_pod_layouts_match_the_header():
    assert C.sizeof(api.PtPrim) == 104 and C.sizeof(api.PtNode) == 56 and C.sizeof(api.PtMaterial) == 16
    assert C.sizeof(api.PtTexture) == 48 and C.sizeof(api.PtCamera) == 152 and C.sizeof(api.PtHitOut) == 40


def test_pod_layouts_cover_the_stats_struct():
    # PtStats as the header declares it: 6 x 8-byte head, 11 doubles, 64 counters, tail_from + padding
    assert C.sizeof(api.PtStats) == 8 + 8 + 4 * 4 + 11 * 8 + 64 * 8 + 8
    assert api.PtStats.tail_from.offset == C.sizeof(api.PtStats) - 8


def test_static_archive_links_from_plain_c():
    """include/ptgpu.h compiles as C11 (gcc, not g++) and lib/libptgpu.a links with the line INTEGRATION.md section 1 gives a
    build.rs (archive + amdhip64 + stdc++ + dl): `make all` builds bin/abi_check that way.  Without a GPU the program checks the
    documented refusal and exits 77; with one it checks three hand-derived hits and a frame (tests/test_gpu_round3.py)."""
    import subprocess

    pkg = os.path.join(ROOT, "rust_pathtracer_amd")
    exe = os.path.join(pkg, "bin", "abi_check")
    assert os.path.exists(exe), "bin/abi_check is missing: make -C rust_pathtracer_amd all"
    assert os.path.getmtime(exe) >= os.path.getmtime(os.path.join(pkg, "lib", "libptgpu.a")) - 1.0, "abi_check is older than the archive"
    # the executable carries the library's code itself: it must not depend on libptgpu.so
    needed = subprocess.run(["readelf", "-d", exe], capture_output=True, text=True, timeout=60).stdout
    assert "libptgpu" not in needed and "libamdhip64" in needed
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode in (0, 77), r.stdout + r.stderr
    if api.device_count() == 0:
        assert r.returncode == 77 and "refused as documented" in r.stdout


def test_no_gpu_fails_loudly(pt):
    if pt.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(pt.PtError):
        pt.Context(0)
    s = pt.Scene("CornellBox").set_image(8, 8, 1, 2, (0, 0, 0)).build()
    with pytest.raises(pt.PtError):
        s.render(0)  # the drop-in render has no CPU fallback


def product_tree(s):
    nodes = s.nodes_array()
    entries, boxes = [], []
    stack = [int(s.desc.root)]
    while stack:
        ref = stack.pop()
        if ref & api.PT_REF_LEAF:
            first = ref & api.PT_REF_INDEX_MASK
            entries.extend(range(first, first + (6 if ref & api.PT_REF_GROUP6 else 1)))
        else:
            n = nodes[ref]
            entries.append(-1)
            boxes.append(list(n["bmin"]) + list(n["bmax"]))
            stack.append(int(n["right"]))
            stack.append(int(n["left"]))
    return np.array(entries, np.int32), np.array(boxes)


@pytest.mark.parametrize("name", su.NAMED)
@pytest.mark.parametrize("seed", [1, 2])
def test_named_world_flattens_to_the_oracle_tree(oracle, name, seed):
    s, o = su.named_pair(name, layout=seed, bvh=seed, perlin=seed)
    e, b = product_tree(s)
    e2, b2 = o.tree()
    assert s.desc.n_prims == o.num_prims
    assert np.array_equal(e, e2)       # same shape, primitives in depth-first order
    assert np.array_equal(b, b2)       # node boxes bit for bit (bvh.rs:37-50, geometry.rs bounding boxes)
    assert s.desc.n_nodes == (e < 0).sum()
    st = s.settings()
    assert np.array_equal(s.camera(0.37).as_array(), oracle.scene_camera(name, 0.37, st["width"] / st["height"]))


def test_scene_defaults_match_scene_rs(pt):
    expect = {  # scene.rs:47-59, :176-186, :241-251, :331-341, :470-480, :602-612
        "SphereField": (854, 480, 250, 20, (1, 1, 1), True),
        "TwoSpheres": (854, 480, 250, 20, (1, 1, 1), False),
        "LightTest": (854, 480, 2000, 50, (0, 0, 0), False),
        "CornellBox": (400, 400, 1000, 20, (0, 0, 0), False),
        "TriangleTest": (400, 400, 1000, 20, (0, 0, 0), False),
        "ModelTest": (800, 800, 250, 20, (1, 1, 1), False),
    }
    for name, (w, h, spp, depth, bg, anim) in expect.items():
        st = pt.Scene(name).settings()
        assert (st["width"], st["height"], st["spp"], st["max_bounces"]) == (w, h, spp, depth)
        assert st["background"] == tuple(float(x) for x in bg) and st["animation"] == anim
    st = pt.Scene("SphereField").settings()
    assert st["fps"] == 30.0 and st["duration"] == 10.0
    # overrides flow into the camera's aspect ratio (scene.rs:71-78)
    s = pt.Scene("CornellBox").set_image(1920, 1080, 256, 50, (0, 0, 0))
    cam = s.camera(0.0)
    assert abs(cam.horizontal[0] / cam.vertical[1]) == pytest.approx(1920 / 1080, rel=1e-15)


def test_sphere_field_population(pt):
    # scene.rs:105-138: 22*22 - 11 = 473 cells, each one sphere, or two when hollow glass; plus ground and 4 big ones
    s = pt.Scene("SphereField").set_seeds(1, 1, 1).build()
    prims = s.prims_array()
    assert (prims["type"] == api.PRIM_SPHERE).all()
    hollow = int((prims["d"][:, 3] == (0.2 * -1.0) + 0.02).sum())  # radius.neg() + 0.02, scene.rs:133
    assert len(prims) == 1 + 473 + hollow + 4 and 20 <= hollow <= 80
    assert (prims["d"][:, 3] == -0.95).sum() == 1 and (prims["d"][:, 3] == 1000.0).sum() == 1


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_random_world_flattens_to_the_oracle_tree(oracle, seed):
    s, o = su.random_world(seed, n_objects=30, with_mesh=(seed % 2 == 0))
    e, b = product_tree(s)
    e2, b2 = o.tree()
    assert np.array_equal(e, e2) and np.array_equal(b, b2)
    # the six sides of every AABox are consecutive rectangles in the order XY,XY,XZ,XZ,YZ,YZ (geometry.rs:391-446)
    nodes = s.nodes_array()
    prims = s.prims_array()
    refs = np.concatenate([nodes["left"], nodes["right"]])
    for ref in refs[(refs & api.PT_REF_GROUP6) != 0]:
        first = int(ref & api.PT_REF_INDEX_MASK)
        assert prims["type"][first:first + 6].tolist() == [1, 1, 2, 2, 3, 3]
        assert prims["d"][first:first + 6, 5].tolist() == [-1, 1, -1, 1, -1, 1]


def test_synthetic_mesh_world_matches_oracle(oracle):
    s, o = su.model_pair(40, 21, 1.2, (0.0, 1.5, 0.0))
    assert s.desc.n_prims == 1 + 2 * 40 * 20
    e, b = product_tree(s)
    e2, b2 = o.tree()
    assert np.array_equal(e, e2) and np.array_equal(b, b2)
    # closed and outward-wound: every camera ray that enters the mesh's bounding sphere region hits a front face
    cam = oracle.scene_camera("ModelTest", 0.0, 1.0)
    ro, rd = su.camera_rays(cam, 32, 32, 4000, 3)
    prim, t, n, ff = o.trace(ro, rd)
    mesh_hits = prim >= 1
    assert mesh_hits.mean() > 0.05 and ff[mesh_hits].all()


def test_reference_error_behaviour(pt):
    m = [su._mat(0, 0)[0]]
    t = [su._tex(0)[0]]
    # geometry.rs:197-199: start.z != end.z -> Err(RectangleNotAxisAlignedError), Display "rectangle axis is not aligned"
    with pytest.raises(pt.PtError, match="rectangle axis is not aligned"):
        pt.world_create([su._obj(1, 0, [0, 0, 0, 1, 1, 0.5, 1.0])[0], su._obj(0, 0, [0, 0, 0, 1])[0]], m, t)
    # bvh.rs:76: fewer than two objects panics
    with pytest.raises(pt.PtError, match="creating BVH node from empty object list"):
        pt.world_create([su._obj(0, 0, [0, 0, 0, 1])[0]], m, t)
    with pytest.raises(pt.PtError, match="creating BVH node from empty object list"):
        pt.world_create([], m, t)
    with pytest.raises(pt.PtError, match="unknown scene"):
        pt.Scene("NoSuchScene")
    # direction is signum'd (geometry.rs:203): -2.5 -> -1, 0.5 -> 1, -0.0 -> -1
    s = pt.world_create([su._obj(2, 0, [0, 0, 0, 1, 0, 1, -2.5])[0], su._obj(2, 0, [0, 1, 0, 1, 1, 1, 0.5])[0],
                         su._obj(2, 0, [0, 2, 0, 1, 2, 1, -0.0])[0]], m, t)
    assert sorted(s.prims_array()["d"][:, 5].tolist()) == [-1.0, -1.0, 1.0]
    # start/end are sorted per axis (geometry.rs:201-202)
    s = pt.world_create([su._obj(1, 0, [3, 4, 7, 1, 2, 7, 1.0])[0], su._obj(0, 0, [0, 0, 0, 1])[0]], m, t)
    r = [p for p in s.prims_array() if p["type"] == api.PRIM_RECT_XY][0]
    assert r["d"][:5].tolist() == [7.0, 1.0, 3.0, 2.0, 4.0]


OBJ = """# quad + triangle, two materials
mtllib test.mtl
o first
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
vn 0 0 1
vn 0 0 1
vn 0 0 1
vn 0 0 1
usemtl glass
f 1//1 2//2 3//3 4//4
o second
v 0 0 1
v 1 0 1
v 0 1 1
usemtl shiny
f -3 -2 -1
usemtl plain
f 5/1 6/2 7/3
"""
MTL = """newmtl glass
illum 7
Ni 1.45
Kd 1 1 1
newmtl shiny
illum 5
Kd 0.7 0.6 0.5
Ns 50
newmtl plain
illum 2
Kd 0.1 0.2 0.3
"""


def test_obj_reader(pt, tmp_path, oracle):
    (tmp_path / "test.obj").write_text(OBJ)
    (tmp_path / "test.mtl").write_text(MTL)
    nm, nt, nl = C.c_uint32(), C.c_uint64(), C.c_uint32()
    assert pt.libpthost.pth_obj_probe(str(tmp_path / "test.obj").encode(), C.byref(nm), C.byref(nt), C.byref(nl)) == 0
    assert (nm.value, nt.value, nl.value) == (3, 4, 3)  # quad fan-triangulated; a new model per usemtl / o
    s = pt.Scene("ModelTest").set_model_path(str(tmp_path / "test.obj")).build()
    meshes, mtls = s.meshes(), s.mtls()
    assert [m.material_id for m in meshes] == [0, 1, 2]
    assert [m.n_indices for m in meshes] == [6, 3, 3] and [m.n_normals for m in meshes] == [12, 0, 0]
    assert np.ctypeslib.as_array(meshes[0].indices, shape=(6,)).tolist() == [0, 1, 2, 0, 2, 3]
    assert np.ctypeslib.as_array(meshes[1].positions, shape=(9,)).tolist() == [0, 0, 1, 1, 0, 1, 0, 1, 1]
    assert [m.illum for m in mtls] == [7, 5, 2] and mtls[0].ni == pytest.approx(1.45) and mtls[1].ns == 50.0
    # obj_model.rs:32-46: illum 7 -> Dielectric(Ni), 5 -> Metal(Kd, 1/Ns), else Lambertian(Kd)
    d = s.desc
    mats = [(d.materials[i].type, d.materials[i].param) for i in range(d.n_materials)]
    assert (api.MAT_DIELECTRIC, pytest.approx(1.45)) in mats and (api.MAT_METAL, 1.0 / 50.0) in mats
    # same arrays through the oracle's restatement of obj_model.rs: same tree
    om, ol = su.mesh_to_oracle(meshes, mtls)
    o = oracle.OracleScene.named("ModelTest", 1, 1, 1, om, ol)
    e, b = product_tree(s)
    e2, b2 = o.tree()
    assert np.array_equal(e, e2) and np.array_equal(b, b2)
    # stored normals come from the file verbatim (obj_model.rs:90-95), geometric ones otherwise (geometry.rs:512)
    prims = s.prims_array()
    tri = prims[prims["type"] == api.PRIM_TRIANGLE]
    assert len(tri) == 4 and all(p["d"][9:12].tolist() == [0.0, 0.0, 1.0] for p in tri)
    with pytest.raises(pt.PtError, match="could not open OBJ"):
        pt.Scene("ModelTest").set_model_path(str(tmp_path / "missing.obj"))
    (tmp_path / "nomtl.obj").write_text("mtllib gone.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    with pytest.raises(pt.PtError, match="could not open MTL"):  # obj_model.rs:29 `.unwrap()` on the MTL result
        pt.Scene("ModelTest").set_model_path(str(tmp_path / "nomtl.obj"))


def test_accum_reduce_argument_checks_need_no_gpu(pt):
    """pt_accum_reduce refuses an empty or null context list before it touches RCCL or a device."""
    import ctypes as C

    assert pt.libptgpu.pt_accum_reduce(None, 0) == -1  # PT_ERR_BAD_ARG
    assert b"no contexts" in pt.libptgpu.pt_last_error(None)
    arr = (C.c_void_p * 2)(None, None)
    assert pt.libptgpu.pt_accum_reduce(arr, 2) == -1
    assert b"null context" in pt.libptgpu.pt_last_error(None)
    with pytest.raises(pt.PtError):
        pt.accum_reduce([])


def test_sharding_rules(pt):
    for total in (0, 1, 7, 256, 1024):
        for world in (1, 2, 3, 4, 8):
            ranges = [pt.sample_range_for_rank(total, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and sum(c for _, c in ranges) == total
            for (b0, c0), (b1, _) in zip(ranges, ranges[1:]):
                assert b0 + c0 == b1
            assert max(c for _, c in ranges) - min(c for _, c in ranges) <= 1
    assert pt.sample_range_for_rank(1024, 3, 8) == (384, 128)  # config 4: [128g, 128g+128)
    assert pt.frames_for_rank(120, 3, 8) == list(range(3, 120, 8))  # config 5: frame i on GPU i mod 8
    assert sorted(sum((pt.frames_for_rank(120, r, 8) for r in range(8)), [])) == list(range(120))
    with pytest.raises(ValueError):
        pt.sample_range_for_rank(8, 2, 2)


def read_png(path):
    """Minimal decoder for what write_file produces: 8-bit RGB, filter 0, one or more IDAT chunks."""
    import struct
    import zlib

    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(data):
        n, kind = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        assert zlib.crc32(kind + body) == crc
        chunks.append((kind, body))
        pos += 12 + n
    assert [k for k, _ in chunks][0] == b"IHDR" and chunks[-1] == (b"IEND", b"")
    w, h, depth, colour, comp, filt, interlace = struct.unpack(">IIBBBBB", chunks[0][1])
    assert (depth, colour, comp, filt, interlace) == (8, 2, 0, 0, 0)  # png::ColorType::Rgb, main.rs:87
    raw = zlib.decompress(b"".join(b for k, b in chunks if k == b"IDAT"))
    rows = np.frombuffer(raw, np.uint8).reshape(h, w * 3 + 1)
    assert not rows[:, 0].any()
    return rows[:, 1:].reshape(h, w, 3)


def test_png_writer_round_trip(pt, tmp_path):
    # main.rs:78-91: 8-bit RGB, rows top first
    rng = np.random.default_rng(4)
    for w, h in [(1, 1), (7, 3), (400, 225)]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        path = str(tmp_path / f"a_{w}x{h}.png")
        pt.write_png(path, img)
        assert np.array_equal(read_png(path), img)
        # and an independent decoder (PIL): the container is a PNG any reader accepts, not just the one above
        Image = pytest.importorskip("PIL.Image")
        with Image.open(path) as im:
            assert im.mode == "RGB" and im.size == (w, h)
            assert np.array_equal(np.asarray(im), img)
    with pytest.raises(pt.PtError, match="could not create image file"):
        pt.write_png(str(tmp_path / "no_such_dir" / "a.png"), np.zeros((2, 2, 3), np.uint8))


def test_guide_tree_builder(pt):
    """The ordered walk's guide tree (guide_tree.h), built and verified on the host: every box a leaf exactly once, inside
    every box stored above it -- for scattered boxes, flat ones (rectangles), many identical ones, a long thin row, and
    huge next to tiny."""
    rng = np.random.default_rng(4)
    def boxes(lo, size):
        return np.concatenate([lo, lo + size], axis=1)
    n = 5000
    cases = {
        "scattered": boxes(rng.uniform(-100, 100, (n, 3)), rng.uniform(0.01, 5, (n, 3))),
        "flat": boxes(rng.uniform(-10, 10, (n, 3)), rng.uniform(0, 2, (n, 3)) * (rng.integers(0, 3, (n, 1)) != np.arange(3))),
        "identical": np.tile([[1.0, 2.0, 3.0, 1.5, 2.5, 3.5]], (777, 1)),
        "row": boxes(np.stack([np.arange(n) * 1.0, np.zeros(n), np.zeros(n)], axis=1), np.full((n, 3), 0.5)),
        "scales": boxes(rng.uniform(-1, 1, (n, 3)) * 10.0 ** rng.integers(-6, 6, (n, 1)), 10.0 ** rng.uniform(-8, 5, (n, 3))),
        "two": np.array([[0, 0, 0, 1, 1, 1], [0, 0, 0, 1, 1, 1.0]]),
    }
    for name, b in cases.items():
        nodes, depth = pt.selftest_guide_tree(b)
        assert nodes == len(b) - 1, name
        assert depth <= 64, (name, depth)
    assert pt.selftest_guide_tree(cases["scattered"])[1] <= 30  # surface-area splits keep a scattered set shallow
    for bad in (np.array([[0, 0, 0, 1, 1, np.inf], [0, 0, 0, 1, 1, 1.0]]), np.array([[0, 0, 0, 1, 1, 1.0]]),
                np.array([[0, 0, 0, 1, 1, 1.0], [2, 0, 0, 1, 1, 1.0]]), np.array([[0, 0, 0, 1, 1, 1e38], [0, 0, 0, 1, 1, 1.0]])):
        with pytest.raises(pt.PtError):
            pt.selftest_guide_tree(bad)


def test_obj_reader_edge_cases(pt, tmp_path):
    """CRLF line ends, comments, negative (relative) indices, a five-sided face (fan of three), v/vt/vn and v//vn and bare
    references in one file, a group statement starting a new model."""
    (tmp_path / "e.obj").write_bytes(
        b"mtllib e.mtl\r\n# comment\r\no first\r\nv 0 0 0\r\nv 1 0 0\r\nv 1 1 0\r\nv 0 1 0\r\nv 0.5 1.5 0\r\nvn 0 0 1\r\nvt 0 0\r\n"
        b"usemtl glass\r\nf -5/1/1 -4/1/1 -3/1/1 -2/1/1 -1/1/1\r\ng grp\r\nf 1//1 2//1 3//1\r\nf 1 2 4\r\n")
    (tmp_path / "e.mtl").write_text("newmtl glass\nillum 7\nNi 1.5\nKd 1 1 1\n")
    nm, nt, nl = C.c_uint32(), C.c_uint64(), C.c_uint32()
    assert pt.libpthost.pth_obj_probe(str(tmp_path / "e.obj").encode(), C.byref(nm), C.byref(nt), C.byref(nl)) == 0
    assert (nm.value, nt.value, nl.value) == (2, 5, 1)
    s = pt.Scene("ModelTest").set_model_path(str(tmp_path / "e.obj")).build()
    meshes = s.meshes()
    assert np.ctypeslib.as_array(meshes[0].indices, shape=(9,)).tolist() == [0, 1, 2, 0, 2, 3, 0, 3, 4]  # -5..-1 = vertices 1..5, fanned
    assert np.ctypeslib.as_array(meshes[0].positions, shape=(15,)).tolist() == [0, 0, 0, 1, 0, 0, 1, 1, 0, 0, 1, 0, 0.5, 1.5, 0]
    (tmp_path / "bad.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 9\n")
    with pytest.raises(pt.PtError):
        pt.Scene("ModelTest").set_model_path(str(tmp_path / "bad.obj"))


def test_host_layer_is_clean_under_sanitizers():
    """the host code (scene builders, BvhNode::new, flattening, OBJ/MTL reader, PNG writer, C API error
    paths) built with -fsanitize=address,undefined and driven by tools/host_selfcheck.cpp; any report aborts the run."""
    import shutil
    import subprocess

    if shutil.which("g++") is None:
        pytest.skip("no g++")
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rust_pathtracer_amd")
    r = subprocess.run(["make", "-C", pkg, "asan"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "all checks passed" in r.stdout


def test_triangle_split_pieces_cover_the_triangle(pt):
    """guide_tree.h split_pieces (option "guide_split"): the boxes of the pieces cover every point of the triangle, a
    compact triangle stays whole, a diagonal sliver is cut into boxes far smaller than its own."""
    rng = np.random.default_rng(11)

    def covered(tri, ratio):
        b = pt.selftest_split_triangle(tri, ratio)
        wgt = rng.random((4000, 3)) ** 2
        wgt /= wgt.sum(1, keepdims=True)
        pts = np.concatenate([wgt @ tri, tri, 0.5 * (tri + np.roll(tri, 1, 0))])
        scale = np.abs(tri).max() * 1e-12
        inside = ((pts[:, None, :] >= b[None, :, :3] - scale) & (pts[:, None, :] <= b[None, :, 3:] + scale)).all(2).any(1)
        lo, hi = tri.min(0), tri.max(0)
        assert (b[:, :3] >= lo - scale).all() and (b[:, 3:] <= hi + scale).all()  # no piece sticks out of the triangle's box
        return len(b), bool(inside.all())

    assert covered(np.array([[0.0, 0, 0], [1, 0, 0], [0, 1, 0]]), 4) == (1, True)  # axis-aligned, compact: one reference
    sliver = np.array([[0.0, 10.5, 0.0], [0.62, 10.51, 0.62], [0.6161, 10.51, 0.6239]])
    n, ok = covered(sliver, 4)
    assert ok and 8 <= n <= 32
    b = pt.selftest_split_triangle(sliver, 4)
    area = lambda q: (q[:, 3] - q[:, 0]) * (q[:, 5] - q[:, 2])
    assert area(b).sum() < 0.1 * 0.62 * 0.6239  # the pieces' footprints against the whole box's
    for k in range(300):
        tri = rng.normal(size=(3, 3)) * 10 ** rng.uniform(-3, 3) + rng.normal(size=3) * 10 ** rng.uniform(-2, 4)
        if k % 3 == 0:
            tri[2] = tri[1] + (tri[1] - tri[0]) * 0.01 + rng.normal(size=3) * 1e-3 * np.abs(tri).max()
        if k % 7 == 0:
            tri[2] = tri[0] + (tri[1] - tri[0]) * 0.5  # degenerate: three points on a line
        n, ok = covered(tri, int(rng.integers(1, 9)))
        assert ok and 1 <= n <= 32, (tri, n)
    with pytest.raises(pt.PtError):
        pt.selftest_split_triangle(np.array([[0.0, 0, 0], [1, 0, np.nan], [0, 1, 0]]), 4)


def test_bench_line_splits_rays_between_the_kernels_that_traced_them():
    """bench.py's per-kernel roofline entries: PtStats.tail_from splits rays_per_bounce between the separate trace / shade launches
    and the tail launch, and every kernel's algorithmic bytes (trace 60 B per ray; shade 76 in + 64 per survivor;
    tail 64 in + 64 per survivor) are those of ITS rays.  Pure host arithmetic on a synthetic PtStats."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    st = api.PtStats()
    rpb = [1000, 600, 300, 100, 40, 10, 2]
    for b, v in enumerate(rpb):
        st.rays_per_bounce[b] = v
    st.rays, st.tail_from = sum(rpb), 3
    st.ms_trace, st.ms_shade, st.ms_path = 3.0, 1.5, 2.0
    st.ms_busy_trace, st.ms_busy_shade, st.ms_busy_path = 3.0, 1.5, 2.0
    st.launches_trace, st.launches_shade, st.launches_path = 3, 3, 1
    tot = bench.sum_stats([st, st])  # two render calls of one step
    assert tot["rays_wavefront"] == 2 * 1900 and tot["survivors_wavefront"] == 2 * (600 + 300 + 100)
    assert tot["rays_tail"] == 2 * 152 and tot["survivors_tail"] == 2 * 52 and tot["tail_from"] == 3
    assert tot["rays_wavefront"] + tot["rays_tail"] == tot["rays"]
    k = {e["kernel"]: e for e in bench.kernel_entries(tot, {}, 0.0, 256, 2.4e9)}
    assert k["trace_kernel"]["rays"] == 3800 and k["trace_kernel"]["bytes_per_launch"] == 3800 * 60 // 6
    assert k["shade_kernel"]["bytes_per_launch"] == (3800 * 76 + 2000 * 64) // 6
    assert k["path_kernel"]["rays"] == 304 and k["path_kernel"]["bytes_per_launch"] == (304 * 64 + 104 * 64) // 2
    for e in k.values():  # achieved = bytes / sum of launch durations; frac against 8 TB/s
        want = e["bytes_per_launch"] * e["launches"] / (e["launch_ms_sum"] * 1e-3) / 1e9
        assert abs(e["achieved"] - want) <= 0.01 * max(1.0, want) and abs(e["frac"] - want / 8000.0) < 1e-4
    # no tail launch: everything belongs to trace and shade, and there is no path entry
    st.tail_from, st.ms_path, st.launches_path = 50, 0.0, 0
    tot = bench.sum_stats([st])
    assert tot["rays_wavefront"] == 2052 and tot["rays_tail"] == 0
    assert [e["kernel"] for e in bench.kernel_entries(tot, {}, 0.0, 256, 2.4e9)] == ["trace_kernel", "shade_kernel"]
    # the issue roofline weights each kernel's instructions per ray by the rays it traced
    pmc = {"trace_kernel": {"valu_per_64_rays": 1000.0}, "shade_kernel": {"valu_per_64_rays": 500.0}, "path_kernel": {"valu_per_64_rays": 9000.0}}
    st.tail_from, st.ms_path, st.launches_path = 3, 2.0, 1
    tot = bench.sum_stats([st])
    ir = bench.issue_roofline(pmc, tot, 1000.0, 256, 2.4e9)
    assert abs(ir["valu_per_64_segments"] - (1500.0 * 1900 + 9000.0 * 152) / 2052) < 0.1
