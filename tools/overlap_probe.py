"""Without a profiler attached: does the visual stack (main stream) start while the language stack (side stream) runs?
HIP events around the language stack and around the visual stack's first / last kernels of the forward, eager steps and plan replays."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xlxmert_amd.config import XLxmertConfig
from xlxmert_amd.engine import Engine, reserve_streams
from xlxmert_amd.trainer import PretrainStep, synthetic_batch
reserve_streams("cuda:0")
cfg = XLxmertConfig()
B = 256
PLAN = os.environ.get("PROBE_PLAN", "1") == "1"
tr = PretrainStep(cfg, B, 20, 64, dtype=torch.bfloat16, device="cuda:0", seed=9595, total_steps=1000, plan=PLAN, drop_grads=True,
                  overlap_optimizer=True)
g = torch.Generator().manual_seed(9595)
tr.set_centroids(torch.randn(cfg.num_clusters, cfg.visual_feat_dim, generator=g).relu())
batches = [{k: v.cuda() for k, v in synthetic_batch(cfg, B, 20, 8, seed=9595 + i).items()} for i in range(4)]
E = lambda: torch.cuda.Event(enable_timing=True)
eng = tr.engine
# timing events recorded THROUGH the C ABI (xl_event_record), so that a recorded launch plan contains them
ev = {}
for k in ("l0", "l1", "v0", "v1", "vend", "b0", "b1", "r0", "r1", "xend", "hend", "xbend", "t0", "t1"):
    ev[k] = E(); ev[k].record()
torch.cuda.synchronize()
def mark(k):
    eng.ops.event_record(ev[k].cuda_event, torch.cuda.current_stream())
orig_lang = eng._language_stack_forward
def lang():
    mark("l0")
    orig_lang()
    mark("l1")
eng._language_stack_forward = lang
orig_cg = eng.ops.codebook_gather
def cg(*a, **k):
    mark("v0")
    orig_cg(*a, **k)
    mark("v1")
eng.ops.codebook_gather = cg
orig_join = eng.join
state = {"first_join": True}
def join():
    if state["first_join"]:
        mark("vend")
        state["first_join"] = False
    orig_join()
eng.join = join
orig_pr = eng._pr
def pr(key):
    if key == "heads":
        mark("xend")
    orig_pr(key)
eng._pr = pr
orig_eb = eng._encoder_backward
def eb(*a, **k):
    mark("hend")
    orig_eb(*a, **k)
eng._encoder_backward = eb
orig_ready = eng._ready
def ready(prefix):
    orig_ready(prefix)
    if prefix == "bert.encoder.x_layers.0.":
        mark("xbend")
eng._ready = ready
orig_vmfb = eng.vis_mask_forward_backward
def vmfb(*a, **k):
    mark("t0")
    r = orig_vmfb(*a, **k)
    mark("t1")
    return r
eng.vis_mask_forward_backward = vmfb
orig_embed_bwd = eng.ops.embed_bwd
def embed_bwd(*a, **k):
    orig_embed_bwd(*a, **k)
    mark("b1")
eng.ops.embed_bwd = embed_bwd
sa, ffn = eng.lang_layers[-1]
orig_lffn_bwd = ffn.bwd
def lffn_bwd(*a, **k):
    mark("b0")
    orig_lffn_bwd(*a, **k)
ffn.bwd = lffn_bwd
vsa, vffn = eng.vis_layers[-1]
orig_vffn_bwd = vffn.bwd
def vffn_bwd(*a, **k):
    mark("r0")
    orig_vffn_bwd(*a, **k)
    mark("r1")
vffn.bwd = vffn_bwd
for i in range(10):
    state["first_join"] = True
    s0 = E(); s0.record()
    tr.step(batches[i % 4])
    s1 = E(); s1.record()
    torch.cuda.synchronize()
    if i >= 3:
        f = lambda a, b: ev[a].elapsed_time(ev[b]) * 1e3
        print(f"step {i}: {s0.elapsed_time(s1):6.2f} ms | language stack {f('l0', 'l1'):7.0f} us | visual stack: first kernel begins {f('l0', 'v0'):7.0f} us "
              f"after the language stack began, is done at {f('l0', 'v1'):7.0f} us, relational stack done at {f('l0', 'vend'):7.0f} us"
              f" || phases on the main stream (ms): stacks fwd {f('t0', 'vend') / 1e3:.2f}, cross layers fwd {f('vend', 'xend') / 1e3:.2f}, head fwd+bwd {f('xend', 'hend') / 1e3:.2f}, "
              f"cross layers bwd {f('hend', 'xbend') / 1e3:.2f}, stacks bwd {f('xbend', 't1') / 1e3:.2f}, total fwd+bwd {f('t0', 't1') / 1e3:.2f}"
              f" || backward: language stack {f('b0', 'b1'):7.0f} us; relational stack's first FFN block begins {f('b0', 'r0'):7.0f} us after it began, done at {f('b0', 'r1'):7.0f} us")
