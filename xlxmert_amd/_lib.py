"""ctypes binding of libxlxmert_hip.so.  Signatures are parsed from include/xlxmert_hip.h so that the
Python side cannot drift from the C ABI.  There is NO fallback: if the library cannot be loaded the
import of any compute path raises."""
import ctypes
import os
import re

import torch  # noqa: F401  -- must come first: loads the HIP runtime this process will use (torch bundles its own)

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), "include", "xlxmert_hip.h")
EXPERIMENTAL = os.environ.get("XL_EXPERIMENTAL", "0") not in ("", "0")     # the experimental build (kernel variants that lost inside the step)
LIB_PATH = os.environ.get("XL_LIB", os.path.join(HERE, "libxlxmert_hip_exp.so" if EXPERIMENTAL else "libxlxmert_hip.so"))   # XL_LIB: debug builds

_CTYPES = {"int": ctypes.c_int, "float": ctypes.c_float, "uint64_t": ctypes.c_uint64, "int64_t": ctypes.c_int64}


def parse_header(path=HEADER, experimental=None):
    """-> {name: (restype, [(argtype, argname), ...])} for every `xl_*` prototype in the header.  experimental: None = all,
    False = only the default library's, True = only those inside `#ifdef XL_EXPERIMENTAL`."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    m = re.search(r"#ifdef XL_EXPERIMENTAL(.*?)#endif", src, flags=re.S)
    if experimental is True:
        src = m.group(1) if m else ""
    elif experimental is False and m:
        src = src[:m.start()] + src[m.end():]
    src = re.sub(r"//[^\n]*", "", src)
    protos = {}
    for m in re.finditer(r"(const\s+char\s*\*|int64_t|int)\s+(xl_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        alist = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.*?)(\w+)$", a)
                alist.append((mm.group(1).strip(), mm.group(2)))
        protos[name] = ("char*" if "char" in ret else ret, alist)
    return protos


def _ctype(t):
    if "*" in t:
        return ctypes.c_void_p
    t = t.replace("const", "").strip()
    return _CTYPES[t]


class XlError(RuntimeError):
    pass


class Lib:
    def __init__(self, path=LIB_PATH):
        if not os.path.exists(path):
            raise XlError(f"{path} not found: build it first (python -m xlxmert_amd.build, or __graft_entry__.build()). "
                          "There is no CPU / eager fallback for the X-LXMERT hot path.")
        self._dll = ctypes.CDLL(path)
        self.protos = parse_header(experimental=False)
        exp = parse_header(experimental=True)
        self.experimental = bool(exp) and all(hasattr(self._dll, n) for n in exp)     # the -DXL_EXPERIMENTAL build exports them all
        if self.experimental:
            self.protos.update(exp)
        for name, (ret, args) in self.protos.items():
            try:
                fn = getattr(self._dll, name)
            except AttributeError:
                raise XlError(f"{path} lacks {name}, which include/xlxmert_hip.h declares: the library is older than the header -- "
                              "rebuild it (python -m xlxmert_amd.build; XL_EXPERIMENTAL=1 for the experimental one)") from None
            fn.argtypes = [_ctype(t) for t, _ in args]
            fn.restype = {"char*": ctypes.c_char_p, "int64_t": ctypes.c_int64}.get(ret, ctypes.c_int)
        self.path = path
        # A library built from an older header (two libraries live side by side: the default and the XL_EXPERIMENTAL=1 build, each
        # rebuilt on its own) would take today's argument lists for yesterday's prototypes.  The launch-plan table carries every
        # plan-able entry point's compiled argument count: compare with the header before the first call.
        for name, (_, args) in self.protos.items():
            fid = self._dll.xl_plan_fn_id(name.encode())
            if fid >= 0 and self._dll.xl_plan_fn_nargs(fid) != len(args):
                raise XlError(f"{path} is stale: {name} was compiled with {self._dll.xl_plan_fn_nargs(fid)} arguments, "
                              f"include/xlxmert_hip.h declares {len(args)} -- rebuild it (python -m xlxmert_amd.build"
                              f"{', with XL_EXPERIMENTAL=1' if 'exp' in os.path.basename(path) else ''})")

    def call(self, name, *args):
        """status-returning entry points: 0 = ok, negative = XL_ERR_* (the error convention of include/xlxmert_hip.h).
        Value-returning ones (xl_version, xl_workspace_floats, xl_last_error) go through raw().
        While a recorder is installed (record()), the call is executed AND appended to it."""
        if name not in self.protos:
            raise XlError(f"{name} is an entry point of the experimental build only: XL_EXPERIMENTAL=1 python -m xlxmert_amd.build, "
                          "then run with XL_EXPERIMENTAL=1 (include/xlxmert_hip.h, section EXPERIMENTAL)")
        rc = getattr(self._dll, name)(*args)
        if rc < 0:
            raise XlError(f"{name} failed ({rc}): {self._dll.xl_last_error().decode()}")
        if self.recorder is not None:
            self.recorder.append((name, args))
        return rc

    def raw(self, name):
        return getattr(self._dll, name)

    # ---- launch plans (csrc/plan.hip): record the C-ABI calls of one step, replay them with one call
    recorder = None

    def record(self):
        """context manager: every call() inside is executed and recorded; returns the list of (name, args)."""
        lib = self

        class _Rec:
            def __enter__(self):
                assert lib.recorder is None, "nested recording"
                lib.recorder = []
                return lib.recorder

            def __exit__(self, *a):
                lib.recorder = None

        return _Rec()

    def record_host(self, fn):
        """While recording: mark a point of the step at which the HOST has to act between two C-ABI calls -- `fn()` is run there
        on every replay (a torch.distributed collective of the gradient exchange: it is not an entry point of this library, so
        it cannot be a plan entry; the plan is cut into segments around it).  No-op outside record()."""
        if self.recorder is not None:
            self.recorder.append((HOST_OP, fn))

    def make_plan(self, calls):
        """[(name, args)] -> LaunchPlan.  Arguments become 64-bit words: pointers / integers by value, floats as their bit
        pattern; ctypes arrays (host arrays a call reads at launch) are kept alive by the plan.  A record that contains host
        operations (record_host) becomes a SegmentedPlan: one LaunchPlan per run of C-ABI calls, the host operations between."""
        if any(name is HOST_OP for name, _ in calls):
            return SegmentedPlan(self, calls)
        return LaunchPlan(self, calls)


HOST_OP = object()          # recorder entry (HOST_OP, callable): see Lib.record_host


class SegmentedPlan:
    """A recorded step with host operations inside: [LaunchPlan | callable]...; run() replays them in order.  The data-parallel
    step is the user: ~15 all-reduce issue points cut its ~600 calls into as many segments (one xl_plan_run each)."""

    def __init__(self, lib, calls):
        self.items, seg = [], []
        for name, args in calls:
            if name is HOST_OP:
                if seg:
                    self.items.append(LaunchPlan(lib, seg))
                    seg = []
                self.items.append(args)
            else:
                seg.append((name, args))
        if seg:
            self.items.append(LaunchPlan(lib, seg))
        self.n_calls = sum(it.n_calls for it in self.items if isinstance(it, LaunchPlan))
        self.n_segments = sum(1 for it in self.items if isinstance(it, LaunchPlan))
        self.n_host_ops = len(self.items) - self.n_segments

    def run(self):
        for it in self.items:
            if isinstance(it, LaunchPlan):
                it.run()
            else:
                it()


class LaunchPlan:
    n_segments, n_host_ops = 1, 0

    def __init__(self, lib, calls):
        import struct
        self.lib, self.keep = lib, []
        ids, nargs, words = [], [], []
        for name, args in calls:
            fid = lib._dll.xl_plan_fn_id(name.encode())
            if fid < 0:
                raise XlError(f"{name} cannot be part of a launch plan: {lib._dll.xl_last_error().decode()}")
            types = [t for t, _ in lib.protos[name][1]]
            if len(types) != len(args):
                raise XlError(f"{name}: {len(args)} arguments recorded, prototype has {len(types)}")
            for t, a in zip(types, args):
                if "*" in t:
                    if a is None:
                        w = 0
                    elif isinstance(a, int):
                        w = a
                    else:                       # ctypes array (host memory read at launch): keep it alive with the plan
                        self.keep.append(a)
                        w = ctypes.addressof(a)
                elif t.replace("const", "").strip() == "float":
                    w = struct.unpack("<I", struct.pack("<f", float(a)))[0]
                else:
                    w = int(a) & 0xFFFFFFFFFFFFFFFF
                words.append(w)
            ids.append(fid)
            nargs.append(len(args))
        self.n_calls = len(ids)
        IA, WA = ctypes.c_int * len(ids), ctypes.c_uint64 * max(1, len(words))
        self.handle = lib._dll.xl_plan_create(len(ids), IA(*ids), IA(*nargs), WA(*words))
        if not self.handle:
            raise XlError(f"xl_plan_create failed: {lib._dll.xl_last_error().decode()}")
        self._run = lib._dll.xl_plan_run

    def run(self):
        rc = self._run(self.handle)
        if rc < 0:
            raise XlError(f"xl_plan_run failed ({rc}): {self.lib._dll.xl_last_error().decode()}")

    def __del__(self):
        try:
            self.lib._dll.xl_plan_destroy(self.handle)
        except Exception:
            pass


_LIB = None


def get_lib():
    global _LIB
    if _LIB is None:
        _LIB = Lib()
    return _LIB
