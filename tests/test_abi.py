"""CPU: the C-ABI library builds, loads, and exports every symbol include/cm3_amd.h declares; the
product fails loudly (no CPU fallback) when asked to compute without a GPU.  No compute calls here."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "cm3_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cm3_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from cm3_amd import _lib
    return _lib


def test_header_symbols_all_exported(built):
    declared = _declared_symbols()
    assert len(declared) >= 20
    handle = built.lib()
    for name in declared:
        assert hasattr(handle, name), "libcm3_hip.so does not export %s" % name
    assert sorted(built.SYMBOLS) == declared, "ctypes binding and header disagree"
    assert handle.cm3_abi_version() == built.ABI_VERSION


STRUCTS = {"cm3_particle_desc": "ParticleDesc", "cm3_particle_bufs": "ParticleBufs", "cm3_particle_traj": "ParticleTraj",
           "cm3_checkers_desc": "CheckersDesc", "cm3_checkers_bufs": "CheckersBufs", "cm3_checkers_traj": "CheckersTraj",
           "cm3_actor_particle_desc": "ActorParticleDesc", "cm3_actor_particle_weights": "ActorParticleWeights",
           "cm3_actor_particle_bufs": "ActorParticleBufs", "cm3_actor_checkers_desc": "ActorCheckersDesc",
           "cm3_actor_checkers_weights": "ActorCheckersWeights", "cm3_actor_checkers_bufs": "ActorCheckersBufs",
           "cm3_copy_shift": "CopyShift", "cm3_transition_cols": "TransitionCols", "cm3_row_cols": "RowCols", "cm3_tile_col": "TileCol"}


def test_struct_layouts_match_header(built, tmp_path):
    """sizeof and every field offset of every struct, as a C compiler sees include/cm3_amd.h, against the ctypes mirror."""
    import ctypes
    import subprocess
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "cm3_amd.h"', 'int main(void) {']
    for cname, pyname in STRUCTS.items():
        cls = getattr(built, pyname)
        lines.append('  printf("%s size %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('  printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    got = {}
    for line in subprocess.check_output([str(exe)], text=True).splitlines():
        cname, field, val = line.split()
        got[(cname, field)] = int(val)
    for cname, pyname in STRUCTS.items():
        cls = getattr(built, pyname)
        assert ctypes.sizeof(cls) == got[(cname, "size")], cname
        for fname, _ in cls._fields_:
            assert getattr(cls, fname).offset == got[(cname, fname)], (cname, fname)
    # every struct the header defines is covered
    text = open(os.path.join(ROOT, "include", "cm3_amd.h")).read()
    assert sorted(set(re.findall(r"typedef struct (cm3_[a-z_]+)", text))) == sorted(STRUCTS)


def test_invalid_arguments_return_error_codes_without_a_gpu(built):
    import ctypes
    handle = built.lib()
    d = built.ParticleDesc()
    b = built.ParticleBufs()
    d.n_envs, d.n_agents, d.max_steps = 0, 4, 33
    assert handle.cm3_particle_step_f32(ctypes.byref(d), ctypes.byref(b), None) == -1
    assert b"n_envs" in handle.cm3_last_error()
    d.n_envs, d.n_agents = 16, 9
    assert handle.cm3_particle_step_f32(ctypes.byref(d), ctypes.byref(b), None) == -1
    cd = built.CheckersDesc()
    cb = built.CheckersBufs()
    cd.n_envs, cd.n_agents, cd.n_rows, cd.n_columns, cd.n_obs, cd.max_steps = 8, 2, 4, 8, 2, 33
    assert handle.cm3_checkers_step(ctypes.byref(cd), ctypes.byref(cb), None) == -1
    assert b"odd" in handle.cm3_last_error()


def test_product_has_no_cpu_fallback():
    import torch
    import cm3_amd
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(cm3_amd.Cm3Error):
        cm3_amd.VecParticleEnv(cm3_amd.load_config("particle_stage1"), 1, 0.2, 33, 4, device="cuda:0")
    with pytest.raises(cm3_amd.Cm3Error):
        cm3_amd.VecParticleEnv(cm3_amd.load_config("particle_stage1"), 1, 0.2, 33, 4, device="cpu")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "cm3_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_checkers_step_refuses_batches_beyond_its_32_bit_offsets(built):
    """The fast Checkers step kernel addresses with 32-bit byte offsets; the launcher must refuse a batch whose WIDEST per-env
    record (obs_others from N = 6 on: 16 N (N - 1) bytes, wider than the 75 N-byte window record) would wrap them -- before any
    HIP call, so this runs without a GPU.  (the guard used to look at the window record only.)"""
    import ctypes
    handle = built.lib()
    d = built.CheckersDesc()
    b = built.CheckersBufs()
    fake = 0x10000                       # non-NULL placeholders: never dereferenced on this path
    for name, _ in b._fields_:
        setattr(b, name, fake)
    N = 8
    d.n_agents, d.n_rows, d.n_columns, d.n_obs, d.max_steps = N, 3, 8, 2, 33
    d.grid_stride, d.obs_self_t_stride = 56, 600
    for k in range(N):
        d.agents_r[k], d.agents_c[k] = k % 3, 8 - k // 3      # distinct start cells inside the band
    widest = 16 * N * (N - 1)                                   # 896 B > 600 B
    d.n_envs = (1 << 32) // widest + 1
    assert d.n_envs * 600 < (1 << 32)                           # the window record alone would have passed
    assert handle.cm3_checkers_step(ctypes.byref(d), ctypes.byref(b), None) == -1
    assert b"4 GiB" in handle.cm3_last_error()


def test_library_carries_the_identity_of_its_sources(built):
    """cm3_source_id() = the hash csrc/build.sh computed over the sources = the hash the binding computes over the files next to
    it; the binding refuses a library built from other sources (a green test run against a stale build proves nothing)."""
    import subprocess
    import sys
    handle = built.lib()
    assert handle.cm3_source_id().decode() == built.source_id() and len(built.source_id()) == 16
    code = ("import cm3_amd._lib as l\n"
            "l.source_id = lambda: '0' * 16\n"
            "try:\n    l.lib()\nexcept l.Cm3Error as e:\n    print('REFUSED' if 'other sources' in str(e) else e)\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT,
                         env={k: v for k, v in os.environ.items() if k not in ("CM3_AMD_LIB", "CM3_AMD_ALLOW_STALE")})
    assert "REFUSED" in out.stdout, out.stdout + out.stderr


def test_matrix_kernel_sources_hold_no_inline_assembly():
    """actor.hip / actor_checkers.hip / policy.hip keep the matrix accumulators in architectural VGPRs (CM3_MATRIX_KERNEL: two
    waves per SIMD declared): there an asm statement's registers can be ones a matrix instruction in flight still reads or
    writes, and the compiler's hazard recogniser does not look into asm statements (round 4: an inline `v_max_f32` relu read a
    result before its passes were through; profiles/r04_policy_head.txt (4)).  The rule is enforced here, on the sources.
    (policy.hip also includes particle.hip: its one asm statement, the write-through store, only READS registers that ordinary
    VALU instructions produced.)"""
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cm3_amd", "csrc")
    build = open(os.path.join(root, "build.sh")).read()
    m = re.search(r"for f in ([a-z_ ]+); do\s*\n\s*\"\$\{HIPCC\}\" \$\{FLAGS\} \$\{PHYS\} \$\{MFMA_FORM\}", build)
    assert m, "build.sh: the loop that compiles the matrix translation units was not found"
    units = m.group(1).split()
    assert set(units) == {"actor", "actor_checkers", "policy", "policy_checkers"}
    assert "amdgpu-mfma-vgpr-form=1" not in re.sub(r"#[^\n]*", "", build), "the experimental flag is not part of the default build"
    for name in units + ["actor_common"]:
        path = os.path.join(root, name + (".h" if name == "actor_common" else ".hip"))
        code = re.sub(r"//[^\n]*", "", open(path).read())          # (comments may talk about asm)
        # the one form that is allowed (round 6, policy_checkers.hip): an EMPTY statement whose only operand is a scalar-register
        # pointer -- `asm volatile("" : "+s"(ptr))`, a barrier for loop-invariant code motion that emits no instruction and names no
        # vector register, so it cannot touch a matrix result
        code = re.sub(r'asm\s+volatile\s*\(\s*""\s*:\s*"\+s"\s*\(\w+\)\s*\)', "", code)
        assert not re.search(r"\basm\s*(volatile\s*)?\(", code), "%s: inline assembly in a matrix translation unit" % name
    for name, kernels in (("actor", ["k_actor_particle"]), ("actor_checkers", ["k_ck_actor"]), ("policy", ["k_policy_rollout"])):
        code = open(os.path.join(root, name + ".hip")).read()
        for k in kernels:
            assert re.search(r"__global__ void CM3_MATRIX_KERNEL %s\(" % k, code), k
    # the 512-thread kernels of round 6 declare the same two waves per SIMD themselves (CM3_MATRIX_KERNEL is the 256-thread form)
    for name, kernels in (("actor_checkers", ["k_ck_actor_x3", "k_ck_actor_others_table"]), ("policy_checkers", ["k_ck_policy_rollout"])):
        code = open(os.path.join(root, name + ".hip")).read()
        for k in kernels:
            assert re.search(r"__launch_bounds__\(512\) __attribute__\(\(amdgpu_waves_per_eu\(2\)\)\) %s\(" % k, code), k


def test_device_code_holds_no_packed_float32_cross_half_select():
    """Round 5 (profiles/r05_policy_fault.txt): `v_pk_mul_f32 ... op_sel:[0,1]` returned a wrong low result in lanes 48..63 on the
    chip.  csrc/build.sh builds the physics without SLP vectorisation so that the compiler does not create that form, and
    tools/isa_lint.py checks every device code object of the build for it; here the lint runs on the objects of the library under
    test, and its pattern is checked on the instruction that failed."""
    import glob
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_lint
    assert isa_lint.BAD.search("v_pk_mul_f32 v[80:81], v[84:85], v[80:81] op_sel:[0,1]")
    assert isa_lint.BAD.search("v_pk_add_f32 v[46:47], v[46:47], v[46:47] op_sel:[0,1] op_sel_hi:[1,0]")
    assert not isa_lint.BAD.search("v_pk_mul_f32 v[84:85], v[84:85], v[86:87] op_sel_hi:[1,0]")
    assert not isa_lint.BAD.search("v_pk_fma_f32 v[72:73], v[72:73], s[80:81], v[92:93] op_sel_hi:[1,0,1]")
    build = open(os.path.join(ROOT, "cm3_amd", "csrc", "build.sh")).read()
    assert "-fno-slp-vectorize" in build and "isa_lint.py" in build
    objs = sorted(glob.glob(os.path.join(ROOT, "cm3_amd", "csrc", "_obj", "*.o")))
    if not objs or not os.path.exists(os.path.join(isa_lint.LLVM, "llvm-objdump")):
        pytest.skip("no build objects / no llvm-objdump here (the build itself runs the lint)")
    seen, bad = isa_lint.lint(objs)
    assert seen >= 9 and not bad, bad[:5]


def test_fault_reproducer_still_compiles(tmp_path):
    """tools/probes/pk_opsel_mfma_repro.hip is the record of WHY the physics is built without SLP vectorisation
    (profiles/r05_policy_fault_upstream_note.md): it must keep compiling for gfx950 with the toolchain of the image."""
    import shutil
    import subprocess
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc here")
    src = os.path.join(ROOT, "tools", "probes", "pk_opsel_mfma_repro.hip")
    out = tmp_path / "repro.o"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-c", src, "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert out.stat().st_size > 0 and shutil.which("python") is not None


def test_sanitizer_pass_is_recorded_and_repeatable():
    """SURVEY.md section 5 / VERDICT r5 item 7: the host side of the C ABI under AddressSanitizer + UndefinedBehaviorSanitizer.
    tools/asan_abi.sh builds the library with the sanitizers and runs this file against it (3 minutes on 8 vCPUs: too long for the
    default CPU suite, so it runs here only with CM3_RUN_ASAN=1); profiles/r06_asan_abi.txt is the record of the last pass."""
    import subprocess
    script = os.path.join(ROOT, "tools", "asan_abi.sh")
    assert os.path.exists(script) and "fsanitize=address,undefined" in open(script).read()
    rec = open(os.path.join(ROOT, "profiles", "r06_asan_abi.txt")).read()
    assert "asan: clean" in rec and "passed" in rec
    if os.environ.get("CM3_RUN_ASAN") == "1" and not os.environ.get("CM3_AMD_LIB"):
        out = subprocess.run(["bash", script], capture_output=True, text=True, cwd=ROOT)
        assert out.returncode == 0 and "asan: clean" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
