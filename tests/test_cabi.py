"""CPU-side checks of the drop-in boundary: the library builds for gfx950, loads, and exports every symbol that
include/xlxmert_hip.h declares; the product refuses to run without a GPU (no CPU fallback)."""
import os

import pytest
import torch


def test_library_builds_and_exports_every_declared_symbol():
    from xlxmert_amd.build import build_library
    from xlxmert_amd._lib import Lib, parse_header
    lib = Lib(build_library())
    exp = parse_header(experimental=True)
    protos = parse_header(experimental=None if lib.experimental else False)
    assert len(protos) >= 20 and len(exp) >= 7 and "xl_set_gemm_relay" in exp and "xl_gemm_pair" in exp
    for name in protos:
        assert hasattr(lib._dll, name), name
    if not lib.experimental:
        # the default library contains neither the experimental kernels nor their entry points (include/xlxmert_hip.h, EXPERIMENTAL)
        for name in exp:
            assert not hasattr(lib._dll, name), name
        import subprocess
        syms = subprocess.run(["nm", "-D", "--defined-only", lib.path], capture_output=True, text=True).stdout
        for frag in ("relay_kernel", "gemm_bf16_q_kernel", "pp_persist_kernel", "pp_pair_kernel"):
            assert frag not in syms, frag
    assert lib.raw("xl_version")() == 1
    must = {"xl_gemm", "xl_layernorm_fwd", "xl_layernorm_bwd", "xl_sdpa_fwd", "xl_sdpa_bwd", "xl_embed_ln_fwd",
            "xl_codebook_gather", "xl_ce_fwd_bwd", "xl_featloss_fwd_bwd", "xl_adamw", "xl_sumsq", "xl_last_error"}
    assert must <= set(protos)


def test_argument_validation_without_gpu():
    """bad arguments are rejected on the host before any launch (error text via xl_last_error)."""
    from xlxmert_amd._lib import XlError, get_lib
    lib = get_lib()
    with pytest.raises(XlError, match="bad shape"):
        lib.call("xl_gemm", None, None, None, None, None, None, 0, 4, 4, 4, 4, 4, 0, 0, 1, 1, 1, 1, 0, 1.0, 0, 0.0, 0, None, None, None)
    with pytest.raises(XlError, match="nq,nk"):
        lib.call("xl_sdpa_fwd", 16, 16, 16, None, 16, 16, 1, 1, 513, 8, 64, 64, 64, 64, 64, 1.0, 0.0, 0, None, None, 0, 0, None, 1, None)   # (<= 512: long kernels)
    # saved dropout decisions only exist on the on-chip bf16 kernels: 0 bytes and a refusal everywhere else
    kb = lib.raw("xl_sdpa_keep_bits_bytes")
    assert kb(256, 12, 64, 64, 64, 1) == 256 * 12 * 2 * 2 * 32 * 4 and kb(256, 12, 20, 64, 64, 1) == 256 * 12 * 1 * 2 * 32 * 4
    assert kb(256, 12, 65, 64, 64, 1) == 0 and kb(256, 12, 64, 64, 64, 0) == 0 and kb(256, 12, 64, 64, 48, 1) == 0
    with pytest.raises(XlError, match="keep_bits"):
        lib.call("xl_sdpa_fwd", 16, 16, 16, None, 16, 16, 1, 1, 64, 64, 64, 64, 64, 64, 64, 1.0, 0.1, 0, None, None, 0, 0, 16, 0, None)     # fp32
    with pytest.raises(XlError, match="keep_bits"):
        lib.call("xl_sdpa_bwd", 16, 16, 16, None, 16, 16, 16, 16, 16, 1, 1, 100, 64, 64, 64, 64, 64, 64, 64, 64, 64, 1.0, 0.1, 0, None, None,
                 None, None, 0, 0, 16, 1, None)                                                                                        # long


def test_a_library_older_than_the_header_is_refused(tmp_path):
    """two libraries live side by side (default / XL_EXPERIMENTAL=1) and each is rebuilt on its own: one that predates a prototype
    change must be refused at load time, not called with today's argument lists.  Stand-in for the stale library: today's library
    against a copy of the header in which one plan-able prototype has grown an argument."""
    from xlxmert_amd import _lib
    hdr = open(_lib.HEADER).read()
    grown = hdr.replace("int xl_colsum(", "int xl_colsum(int one_more, ", 1)
    assert grown != hdr
    fake = tmp_path / "xlxmert_hip.h"
    fake.write_text(grown)
    real_parse = _lib.parse_header
    try:
        _lib.parse_header = lambda path=str(fake), experimental=None: real_parse(str(fake), experimental)
        with pytest.raises(_lib.XlError, match="stale.*xl_colsum"):
            _lib.Lib(_lib.LIB_PATH)
    finally:
        _lib.parse_header = real_parse


def test_no_cpu_fallback():
    from xlxmert_amd._lib import XlError
    from xlxmert_amd.ops import HipOps
    ops = HipOps(torch.float32)
    x = torch.zeros(4, 8)
    with pytest.raises(XlError, match="CPU tensor"):
        ops.colsum(x, torch.zeros(8), 4, 8, 8)


def test_product_package_does_not_import_oracle():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "xlxmert_amd")
    for fn in os.listdir(root):
        if fn.endswith(".py"):
            src = open(os.path.join(root, fn)).read()
            for line in src.splitlines():
                ls = line.strip()
                if ls.startswith("import ") or ls.startswith("from "):
                    assert "oracle" not in ls and "fake_ops" not in ls and "transformers" not in ls, (fn, ls)


def test_launch_plan_records_and_replays_host_side():
    """plan machinery without a GPU: calls are recorded with their argument words (pointers, ints, floats, uint64), replayed
    in order by xl_plan_run, a failing call stops the replay with its own error text, and only plan-able entry points are
    accepted (csrc/plan.hip)."""
    from xlxmert_amd._lib import XlError, get_lib
    lib = get_lib()
    with lib.record() as calls:
        lib.call("xl_set_deferred_reduce", 1)
        lib.call("xl_set_step_seed_ptr", None)
        lib.call("xl_set_deferred_reduce", 0)
    assert [c[0] for c in calls] == ["xl_set_deferred_reduce", "xl_set_step_seed_ptr", "xl_set_deferred_reduce"]
    assert lib.recorder is None
    plan = lib.make_plan(calls)
    assert plan.n_calls == 3
    plan.run()
    plan.run()
    # argument marshalling: xl_gemm validates on the host before any launch -- the words must arrive in their slots
    bad_shape = ("xl_gemm", (16, 16, 16, None, None, None, 0, 4, 4, 4, 4, 4, 0, 0, 1, 1, 1, 1, 0, 1.0, 0, 0.0, 0, None, None, None))
    with pytest.raises(XlError, match="bad shape M=0 N=4 K=4"):
        lib.make_plan([calls[0], bad_shape]).run()
    bad_drop = ("xl_gemm", (16, 16, 16, None, None, None, 8, 4, 4, 4, 4, 4, 0, 0, 1, 1, 1, 1, 0, 1.0, 0, 1.5, 7, None, None, None))
    with pytest.raises(XlError, match="p_drop 1.5"):
        lib.make_plan([bad_drop]).run()
    with pytest.raises(XlError, match="cannot be part of a launch plan"):
        lib.make_plan([("xl_gemm_trace", (None,))])


def test_comm_entry_points_and_rccl_constants():
    """xl_comm_* (csrc/comm.hip) binds RCCL at run time by symbol name and passes ncclDataType_t / ncclRedOp_t by VALUE: the
    values it hard-codes must be the ones of the installed rccl.h; an unknown communicator is an error, not a crash."""
    import re
    from xlxmert_amd._lib import XlError, get_lib, parse_header
    protos = parse_header()
    for fn in ("xl_comm_unique_id", "xl_comm_init", "xl_comm_destroy", "xl_comm_allreduce", "xl_comm_reduce_scatter",
               "xl_comm_allgather", "xl_comm_bcast", "xl_comm_reduce", "xl_comm_wait"):
        assert fn in protos, fn
    hdr = "/opt/rocm/include/rccl/rccl.h"
    if os.path.exists(hdr):
        txt = open(hdr).read()
        want = {"ncclUint8": 1, "ncclInt64": 4, "ncclFloat32": 7, "ncclBfloat16": 9, "ncclSum": 0, "ncclMax": 2}
        for name, val in want.items():
            m = re.search(rf"\b{name}\s*=\s*(\d+)", txt)
            assert m and int(m.group(1)) == val, (name, m and m.group(1))
        assert re.search(r"#define\s+NCCL_UNIQUE_ID_BYTES\s+128", txt)
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "xlxmert_amd", "csrc", "comm.hip")).read()
    assert "NCCL_FLOAT32 = 7" in src and "NCCL_BFLOAT16 = 9" in src and "NCCL_SUM = 0" in src
    lib = get_lib()
    with pytest.raises(XlError, match="unknown communicator|cannot load"):
        lib.call("xl_comm_wait", 12345, None)
