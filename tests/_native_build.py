"""Builds the host-side emulators under tests/native/ with hipcc.

Only the HOST half of those files ever runs (the kernels' __host__ __device__ lane code on the CPU), so the device half is not compiled:
`--cuda-host-only` takes seconds instead of the 90 s a gfx950 code object of the plan kernels costs.  A host object of a file with
__global__ kernels still registers a fat binary at start-up; a stub of eight zero bytes under the symbol it asks for satisfies the linker,
and the runtime never looks at it because no kernel is launched.  Falls back to the full build when anything of that fails."""
import os
import re
import shutil
import subprocess

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O1", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-Wno-pass-failed", "-Wno-inline-asm", "-Wno-unused-result"]


def hipcc_path():
    return HIPCC if os.path.exists(HIPCC) else shutil.which("hipcc")


SANITIZE = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g"]


def build(src: str, exe: str, timeout: int = 900, sanitize: bool = False, extra=()) -> None:
    """sanitize: AddressSanitizer + UndefinedBehaviorSanitizer (clang's runtimes ship with ROCm's llvm); any finding aborts the program.
    extra: more compiler flags (e.g. -DBEVW_UNIT_STORE16=2 for the emulator of a variant build)."""
    hipcc = hipcc_path()
    obj, stub_c, stub_o = exe + ".o", exe + "_fatbin_stub.c", exe + "_fatbin_stub.o"
    sanitize = sanitize or os.environ.get("BEVW_NATIVE_SANITIZE") == "1"   # the sanitizer leg re-runs the emulator tests this way
    san = SANITIZE if sanitize else []
    try:
        r = subprocess.run([hipcc] + FLAGS + list(extra) + san + ["--cuda-host-only", "-c", src, "-o", obj], capture_output=True, text=True, timeout=timeout)
        if r.returncode == 0:
            syms = subprocess.run(["nm", obj], capture_output=True, text=True, timeout=60).stdout
            wanted = sorted(set(re.findall(r"^\s+U (__hip_fatbin\w*)$", syms, flags=re.M)))
            with open(stub_c, "w") as f:
                for s in wanted:
                    f.write('__attribute__((section(".hip_fatbin"), aligned(4096))) const char %s[8] = {0};\n' % s)
                f.write("int bevw_native_build_stub;\n")
            ok = subprocess.run(["gcc", "-c", stub_c, "-o", stub_o], capture_output=True, text=True, timeout=60).returncode == 0
            ok = ok and subprocess.run([hipcc] + san + [obj, stub_o, "-o", exe], capture_output=True, text=True, timeout=300).returncode == 0
            if ok and subprocess.run([exe, "--bevw-selfcheck-noop"], capture_output=True, timeout=120).returncode is not None:
                return
    except (OSError, subprocess.SubprocessError):
        pass
    assert not sanitize, "sanitized host build failed: " + (r.stderr[-2000:] if "r" in dir() else "")
    r = subprocess.run([hipcc] + FLAGS + list(extra) + [src, "-o", exe], capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-2000:]
