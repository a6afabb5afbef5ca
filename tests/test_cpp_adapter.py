"""The C++ drop-in adapter (include/lccrf_densecrf.hpp): the reference's call site
(src/Tracking.cc:1919-1930) compiled with only the two type names changed."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pkg = importlib.import_module("lc-crf-slam_amd")


@pytest.fixture(scope="module")
def exe(tmp_path_factory, po):
    if not os.path.exists(pkg.LIB_PATH):
        pkg.build_library()
    out = str(tmp_path_factory.mktemp("cpp") / "call_site_test")
    cmd = ["g++", "-std=c++14", "-O2", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "call_site_test.cpp"), "-o", out,
           pkg.LIB_PATH, po.ORACLE_SO,
           "-Wl,-rpath," + os.path.dirname(pkg.LIB_PATH), "-Wl,-rpath," + os.path.dirname(po.ORACLE_SO),
           "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True)
    return out


def write_inputs(path, wl, N, seed):
    fr = wl.slam_frame(N, seed)
    with open(path, "wb") as f:
        f.write(np.int32(N).tobytes())
        for a in (fr["obs"], fr["err"], fr["uv"], fr["init_label"]):
            f.write(np.ascontiguousarray(a).tobytes())


def test_adapter_compiles_and_fails_loudly_without_gpu(exe, wl, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = str(tmp_path / "in.bin")
    write_inputs(p, wl, 64, 1)
    r = subprocess.run([exe, p], capture_output=True, text=True)
    assert r.returncode == 3 and "no HIP device" in r.stdout      # throws; no CPU fallback


@pytest.mark.gpu
@pytest.mark.parametrize("N", [2000, 1234, 5])
def test_call_site_matches_oracle(exe, wl, tmp_path, N):
    p = str(tmp_path / "in.bin")
    write_inputs(p, wl, N, 9)
    r = subprocess.run([exe, p], capture_output=True, text=True)
    assert r.returncode == 0 and "CALL-SITE OK" in r.stdout, r.stdout + r.stderr


@pytest.fixture(scope="module")
def adapter_exe(tmp_path_factory, po):
    if not os.path.exists(pkg.LIB_PATH):
        pkg.build_library()
    out = str(tmp_path_factory.mktemp("cpp2") / "adapter_test")
    cmd = ["g++", "-std=c++14", "-O2", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "adapter_test.cpp"), "-o", out,
           pkg.LIB_PATH, po.ORACLE_SO,
           "-Wl,-rpath," + os.path.dirname(pkg.LIB_PATH), "-Wl,-rpath," + os.path.dirname(po.ORACLE_SO),
           "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True)
    return out


def test_adapter_derives_from_the_reference_bases_where_they_exist(tmp_path):
    """With the reference tree on the include path the adapter's base classes ARE the reference's
    (densecrf_base.h): compile the adapter test's translation unit against them (syntax + type check only:
    /root/reference does not exist on the GPU box, and nothing of it is linked)."""
    inc = "/root/reference/Thirdparty/DenseCRF/include"
    if not os.path.isdir(inc):
        pytest.skip("reference tree not present")
    probe = tmp_path / "probe.cpp"
    probe.write_text('#include "lccrf_densecrf.hpp"\n'
                     '#ifndef LCCRF_HAVE_REFERENCE_BASES\n#error "reference bases not picked up"\n#endif\n'
                     '#include "densecrf3d.h"\n'
                     'static_assert(std::is_base_of<DenseCRF::DenseCRF, DenseCRF::DenseCRFHIP<2>>::value, "");\n'
                     'static_assert(std::is_base_of<DenseCRF::PairwisePotential, DenseCRF::PottsPotentialHIP<2, 2>>::value, "");\n'
                     '// one of ours inside the reference\'s own CPU CRF\n'
                     'void f(DenseCRF::DenseCRF3D<2> &cpu, const float *feat, int N) { cpu.addPairwiseEnergy(new DenseCRF::PottsPotentialHIP<2, 2>(feat, N, 3.0f)); }\n')
    for src in (str(probe), os.path.join(ROOT, "tests", "cpp", "adapter_test.cpp")):
        subprocess.run(["g++", "-std=c++14", "-fsyntax-only", "-include", "type_traits", "-I" + os.path.join(ROOT, "include"),
                        "-I" + inc, src], check=True)


@pytest.mark.gpu
@pytest.mark.parametrize("N", [1500, 77])
def test_adapter_base_pointers_apply_and_mixed_potentials(adapter_exe, wl, tmp_path, N):
    p = str(tmp_path / "in.bin")
    write_inputs(p, wl, N, 11)
    r = subprocess.run([adapter_exe, p], capture_output=True, text=True)
    assert r.returncode == 0 and "ADAPTER OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_image_demo_through_the_adapter_gives_the_known_answer(tmp_path, golden):
    """Thirdparty/DenseCRF/examples/example_cpu.cpp with the two type names changed (tests/cpp/image_demo_test.cpp): 21 classes, a 2-D
    and a 5-D potential on 76 800 pixels through DenseCRFHIP<21> -- the labels must colour to res1_cpu.ppm byte for byte."""
    if not os.path.exists(pkg.LIB_PATH):
        pkg.build_library()
    exe = str(tmp_path / "image_demo_test")
    subprocess.run(["g++", "-std=c++14", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "image_demo_test.cpp"),
                    "-o", exe, pkg.LIB_PATH, "-Wl,-rpath," + os.path.dirname(pkg.LIB_PATH), "-Wl,-rpath,/opt/rocm/lib"], check=True)
    z = golden["example_im1"]
    im, res, lab, colors = z["im"], z["res"], z["label"], z["colors"]
    H, W, _ = im.shape
    src, dst = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(src, "wb") as f:
        f.write(np.int32(W).tobytes() + np.int32(H).tobytes())
        f.write(np.ascontiguousarray(im, np.uint8).tobytes())
        f.write(np.ascontiguousarray(lab, np.int16).tobytes())
    r = subprocess.run([exe, src, dst], capture_output=True, text=True)
    assert r.returncode == 0 and "IMAGE DEMO OK" in r.stdout, r.stdout + r.stderr
    m = np.fromfile(dst, np.int16)
    col = colors[m]
    out = np.stack([col & 255, (col >> 8) & 255, (col >> 16) & 255], -1).astype(np.uint8).reshape(H, W, 3)
    assert np.array_equal(out, res)
