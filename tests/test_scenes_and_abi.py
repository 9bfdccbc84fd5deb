"""Host logic: scene generators, struct layouts against the C header, exported symbols of the HIP library."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import numpy as np

import lprt_amd
from lprt_amd import abi, scenes, renderer, dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "lprt.h")


def test_cornell_box_matches_the_survey_spec(cornell):
    assert cornell.n_tris == 36 and len(cornell.objects) == 8 and len(cornell.materials) == 6
    assert len(cornell.lights) == 1 and cornell.lights[0]["type"] == abi.LIGHT_POINT
    assert np.abs(cornell.vertices["position"]).max() <= 1.0 + 1e-6  # local coordinates, fp16-safe
    assert np.allclose(np.linalg.norm(cornell.vertices["normal"], axis=1), 1.0, atol=1e-6)
    t, n = cornell.vertices["tangent"], cornell.vertices["normal"]
    assert np.allclose((t * n).sum(axis=1), 0.0, atol=1e-6)  # synthesised tangents are perpendicular (loader.cu:376-389)
    for o, a in zip(cornell.objects, cornell.aabbs):
        assert np.allclose(o["L2W"].astype(np.float64) @ o["W2L"].astype(np.float64), np.eye(4), atol=1e-5)
        assert (a[:3] <= a[3:]).all()


def test_sponza_like_matches_the_survey_spec(sponza):
    assert abs(sponza.n_tris - 262144) <= 2621 and len(sponza.vaos) == 24 and len(sponza.materials) == 12
    assert [l["type"] for l in sponza.lights] == [abi.LIGHT_DIRECTIONAL, abi.LIGHT_POINT, abi.LIGHT_POINT]
    ends = sponza.vaos[:, 2] + sponza.vaos[:, 3]
    assert (sponza.vaos[1:, 2] == ends[:-1]).all() and ends[-1] == len(sponza.indices)
    assert sponza.indices.max() < len(sponza.vertices)
    again = scenes.sponza_like()
    assert np.array_equal(again.indices, sponza.indices) and np.array_equal(again.vertices, sponza.vertices)


def test_ctypes_structs_match_the_c_header():
    fields = {
        "lprt_vertex": abi.Vertex, "lprt_mesh_vao": abi.MeshVAO, "lprt_material": abi.Material, "lprt_object": abi.Object,
        "lprt_aabb": abi.AABB, "lprt_light": abi.Light, "lprt_camera": abi.Camera, "lprt_skybox": abi.Skybox,
        "lprt_frame_settings": abi.FrameSettings, "lprt_ray": abi.Ray, "lprt_hit": abi.Hit, "lprt_counters": abi.Counters,
    }
    lines = ["#include <stdio.h>", "#include <stddef.h>", f'#include "{HEADER}"', "int main(void){"]
    for cname, cls in fields.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['printf("LPRT_BUF_COUNT %d\\n", (int)LPRT_BUF_COUNT);', 'printf("LPRT_STAGE_COUNT %d\\n", (int)LPRT_STAGE_COUNT);',
              "return 0;}"]
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "t.c"), os.path.join(d, "t")
        open(src, "w").write("\n".join(lines))
        subprocess.run(["gcc", "-std=c99", "-o", exe, src], check=True)
        out = dict(l.split() for l in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in fields.items():
        assert int(out[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(out[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"
    assert int(out["LPRT_BUF_COUNT"]) == len(abi.BUF) and int(out["LPRT_STAGE_COUNT"]) == len(abi.STAGES)


def test_library_loads_and_exports_every_declared_symbol():
    declared = set(re.findall(r"\b(lprt_[a-z0-9_]+)\s*\(", open(HEADER).read()))
    assert declared, "no declarations parsed"
    lib = renderer.load_library()  # raises if any bound symbol is missing
    bound = {name for name, _, _ in renderer.SYMBOLS}
    assert declared == bound, (declared - bound, bound - declared)
    for name in declared:
        getattr(lib, name)
    assert b"gfx950" in lib.lprt_version()


def test_abi_info_matches_the_binding_and_option_names_follow_the_header():
    """lprt_get_abi_info: what renderer.check_abi compares at load time (struct sizes, stage / buffer / option counts,
    words per RNG state) -- and the binding's option names in the header's enum order."""
    lib = renderer.load_library()
    info = abi.AbiInfo()
    lib.lprt_get_abi_info(C.byref(info), C.sizeof(info))
    for k, v in abi.expected_abi_info().items():
        assert getattr(info, k) == v, k
    # a caller that knows a shorter struct gets only that many bytes
    short = (C.c_uint32 * 6)(*([0xdeadbeef] * 6))
    lib.lprt_get_abi_info(short, 16)
    assert short[1] == len(abi.STAGES) and short[4] == 0xdeadbeef
    names = re.findall(r"LPRT_OPT_([A-Z_]+)\s*=\s*(\d+)", open(HEADER).read())
    assert [n.lower() for n, _ in names if n != "COUNT"] == abi.OPTIONS
    assert [int(v) for _, v in names] == list(range(len(names)))
    assert C.sizeof(abi.FrameSettings) == info.sizeof_frame_settings


def test_guard_paged_out_arrays_fault_at_the_first_byte_past_the_end():
    """tests/guarded.py: the arrays the bindings hand to native code in this test suite end at a PROT_NONE page."""
    import subprocess
    import sys
    import guarded
    a = guarded.empty((3, 5), np.float32)
    assert a.shape == (3, 5) and (a.ctypes.data + a.nbytes) % 4096 == 0 and not a.any()
    a[...] = 7
    code = ("import sys, ctypes, numpy as np; sys.path.insert(0, %r); import guarded; a = guarded.empty(100, np.float32); "
            "ctypes.memset(a.ctypes.data, 0, a.nbytes); sys.stdout.write('inside ok'); sys.stdout.flush(); "
            "ctypes.memset(a.ctypes.data, 0, a.nbytes + 1)") % os.path.dirname(os.path.abspath(__file__))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert p.stdout == "inside ok" and p.returncode == -11  # SIGSEGV at the overrun, not later
    assert renderer.alloc_out is guarded.empty  # installed by conftest.py for the whole suite


def test_create_fails_loudly_without_a_gpu_or_with_bad_arguments():
    lib = renderer.load_library()
    h = C.c_void_p()
    assert lib.lprt_create(C.byref(h), 0, 0, 64, 0, 64, 0) == -1  # LPRT_ERR_INVALID: zero width
    assert lib.lprt_create(C.byref(h), 0, 64, 64, 10, 5, 0) == -1
    import torch
    if not torch.cuda.is_available():
        assert lib.lprt_create(C.byref(h), 0, 64, 64, 0, 64, 0) == -3  # LPRT_ERR_NO_DEVICE: no CPU fallback
        assert b"no CPU fallback" in lib.lprt_last_error(None)


def test_strip_partition_covers_the_frame():
    for H, N in [(1080, 1), (1080, 2), (1080, 4), (1080, 8), (2160, 8), (1081, 4), (7, 8)]:
        rows = [dist.strip_rows(H, N, r) for r in range(N)]
        assert rows[0][0] == 0 and rows[-1][1] == H
        assert all(rows[i][1] == rows[i + 1][0] for i in range(N - 1))
        for r in range(N):
            wb, we = dist.window_rows(H, N, r)
            assert 0 <= wb <= rows[r][0] and rows[r][1] <= we <= H
            assert rows[r][0] - wb <= dist.HALO_ROWS and we - rows[r][1] <= dist.HALO_ROWS
    assert dist.strip_rows(1080, 8, 3) == (405, 540) and dist.strip_rows(2160, 8, 7) == (1890, 2160)


def test_image_writer_roundtrip(tmp_path):
    from lprt_amd import image_io
    from PIL import Image
    a = np.random.default_rng(0).uniform(-0.2, 1.3, size=(5, 7, 4)).astype(np.float32)
    a[0, 0, 0] = np.nan
    image_io.save_png(str(tmp_path / "a.png"), a)
    png = np.asarray(Image.open(str(tmp_path / "a.png")))
    assert png.shape == (5, 7, 3) and png[4, 0, 0] == 0                      # row 0 of the frame is the bottom PNG row; NaN -> 0
    assert png[4, 1, 1] == int(np.clip(a[0, 1, 1], 0, 1) * 255 + 0.5)
    image_io.save_pfm(str(tmp_path / "a.pfm"), np.nan_to_num(a))
    assert np.array_equal(image_io.load_pfm(str(tmp_path / "a.pfm")), np.nan_to_num(a)[..., :3])


def test_compat_header_and_demo_build_and_fail_loudly_without_gpu(tmp_path):
    """include/lprt_compat.hpp (reference-named host types over the C-ABI) compiles as plain C++17, and the example
    host program reports the library's error instead of producing a frame when no GPU is there."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "t.cpp"
    src.write_text('#include "lprt_compat.hpp"\nusing namespace lprt_compat;\n'
                   'static_assert(sizeof(RDVertex<float16>) == 64 && sizeof(CUDAMeshVAO) == 16, "layouts");\n'
                   'template class Renderer<float>;\ntemplate class Renderer<float16>;\nint main() { return 0; }\n')
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", os.path.join(root, "include"), str(src)], check=True)
    subprocess.run(["make", "-C", os.path.join(root, "examples"), "compat_demo"], check=True, capture_output=True)
    from lprt_amd import scenes
    blob = scenes.write_blob(scenes.cornell_box(), 64, 48, str(tmp_path / "c.blob"))
    import torch
    if not torch.cuda.is_available():
        p = subprocess.run([os.path.join(root, "examples", "compat_demo"), blob, "64", "48", "1", "1", "f32", str(tmp_path / "o.raw")],
                           capture_output=True, text=True)
        assert p.returncode == 4 and "lprt_create" in p.stderr and not (tmp_path / "o.raw").exists()
