"""The C-ABI boundary from a host that is neither Python nor torch: tools/c_host/unet_step_demo.c is compiled with
gcc (plain C99) against include/commonscenes_hip.h + libcommonscenes_hip.so and run on the MI355X (plan, pack, context,
cs_unet_step, fused DDIM update; determinism, guidance-pair == duplicated batch, workspace check)."""
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.gpu
@pytest.mark.parametrize("width,objects", [(32, 2), (224, 2)])
def test_c_host_drives_cs_unet_step(tmp_path, width, objects):
    from commonscenes_amd import build
    gcc = shutil.which("gcc")
    rocm = Path("/opt/rocm")
    if gcc is None or not (rocm / "include" / "hip" / "hip_runtime_api.h").exists():
        pytest.skip("needs gcc and the ROCm headers")
    lib = build.build_native(verbose=False)
    exe = tmp_path / "unet_step_demo"
    # plain C99: the header must be consumable without a C++ or HIP compiler
    subprocess.run([gcc, "-std=c99", "-O2", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", str(rocm / "include"),
                    "-I", str(ROOT / "include"), str(ROOT / "tools" / "c_host" / "unet_step_demo.c"),
                    "-L", str(lib.parent), "-lcommonscenes_hip", "-L", str(rocm / "lib"), "-lamdhip64", "-lm",
                    f"-Wl,-rpath,{lib.parent}", f"-Wl,-rpath,{rocm / 'lib'}", "-o", str(exe)],
                   check=True, capture_output=True, timeout=300)
    r = subprocess.run([str(exe), str(width), str(objects)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ok: width" in r.stdout and "guidance-pair entry vs duplicated batch" in r.stdout
    assert f"decode: {objects} x 64^3 SDF" in r.stdout
