// Launch plans: a whole training step (~560 kernel launches + ~150 stream hand-offs on four HIP streams) recorded once as a
// list of C-ABI calls and replayed by ONE call from the host language.  The reference trainer pays a Python -> ATen dispatch
// per operator (ref x-lxmert/src/pretrain/lxmert_pretrain.py:334-345: one autograd forward/backward per step); this path's
// Python engine paid 13 ms per step for its ctypes calls against a 20 ms GPU step.  A hipGraph of the same step was
// measured and rejected on ROCm 7.0: hipGraphLaunch spends 15.5 ms of host time on the 560-node, four-branch graph (more than
// the eager enqueue) and the replay runs 22.5 ms against 20.4 ms; stream capture also crashes on a forked stream joined by
// another forked stream (tools/graph_probe2.py).  A plan keeps the eager launch sequence -- same kernels, same streams, same
// events -- and removes the interpreter from it.
//
// Everything that changes from step to step lives in device memory (inputs in the engine's static buffers, the dropout step
// seed behind xl_set_step_seed_ptr, schedule scalars behind xl_schedule_step, the masked-row list padded to a fixed granule),
// so the recorded argument words are replayed verbatim.  An entry = the function's index in kFns + its arguments, one
// 64-bit word each (pointers and integers by value, floats as their bit pattern); the typed unpacking below is generated
// from the prototypes in include/xlxmert_hip.h, so a signature change cannot silently skew a plan.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>
#include <utility>
#include <vector>
#include "common.h"

namespace xl {

template <class T>
static inline T unpack_word(uint64_t w) {
    if constexpr (std::is_pointer_v<T>) {
        return reinterpret_cast<T>(static_cast<uintptr_t>(w));
    } else if constexpr (std::is_same_v<T, float>) {
        const uint32_t u = static_cast<uint32_t>(w);
        float f;
        memcpy(&f, &u, sizeof f);
        return f;
    } else {
        static_assert(std::is_integral_v<T>, "plan arguments are pointers, integers or floats");
        return static_cast<T>(w);
    }
}

template <auto F> struct Invoker;
template <class... A, int (*F)(A...)>
struct Invoker<F> {
    static constexpr int N = sizeof...(A);
    template <size_t... I>
    static int call_(const uint64_t* w, std::index_sequence<I...>) { return F(unpack_word<A>(w[I])...); }
    static int call(const uint64_t* w) { return call_(w, std::index_sequence_for<A...>{}); }
};

struct PlanFn { const char* name; int nargs; int (*call)(const uint64_t*); };
#define XL_PLAN_FN(f) {#f, Invoker<&f>::N, &Invoker<&f>::call}
static const PlanFn kFns[] = {
    XL_PLAN_FN(xl_set_step_seed_ptr), XL_PLAN_FN(xl_gemm),
#ifdef XL_EXPERIMENTAL
    XL_PLAN_FN(xl_gemm_pair),
#endif
    XL_PLAN_FN(xl_gemm_wgrad_group), XL_PLAN_FN(xl_layernorm_fwd),
    XL_PLAN_FN(xl_layernorm_bwd), XL_PLAN_FN(xl_visn_ln_fwd), XL_PLAN_FN(xl_visn_ln_bwd), XL_PLAN_FN(xl_set_deferred_reduce),
    XL_PLAN_FN(xl_flush_reductions), XL_PLAN_FN(xl_embed_ln_fwd), XL_PLAN_FN(xl_embed_bwd), XL_PLAN_FN(xl_codebook_gather),
    XL_PLAN_FN(xl_masked_colsum), XL_PLAN_FN(xl_colsum), XL_PLAN_FN(xl_dropout), XL_PLAN_FN(xl_gelu_bwd), XL_PLAN_FN(xl_tanh_bwd),
    XL_PLAN_FN(xl_bce_logits_fwd_bwd), XL_PLAN_FN(xl_sdpa_fwd), XL_PLAN_FN(xl_sdpa_bwd), XL_PLAN_FN(xl_mask_counts),
    XL_PLAN_FN(xl_ce_fwd_bwd), XL_PLAN_FN(xl_featloss_fwd_bwd), XL_PLAN_FN(xl_gather_rows), XL_PLAN_FN(xl_scatter_rows),
    XL_PLAN_FN(xl_gather_labels), XL_PLAN_FN(xl_sumsq), XL_PLAN_FN(xl_schedule_step), XL_PLAN_FN(xl_adamw),
    XL_PLAN_FN(xl_cast_from_f32), XL_PLAN_FN(xl_cast_to_f32), XL_PLAN_FN(xl_take_f32), XL_PLAN_FN(xl_put_f32), XL_PLAN_FN(xl_memset), XL_PLAN_FN(xl_stream_fork),
    XL_PLAN_FN(xl_rowmax_combine), XL_PLAN_FN(xl_event_record), XL_PLAN_FN(xl_stream_wait), XL_PLAN_FN(xl_ctx_bind),
    XL_PLAN_FN(xl_flush_reductions_on), XL_PLAN_FN(xl_comm_allreduce), XL_PLAN_FN(xl_comm_reduce_scatter), XL_PLAN_FN(xl_comm_allgather), XL_PLAN_FN(xl_comm_wait),
};
constexpr int kNumFns = sizeof(kFns) / sizeof(kFns[0]);

struct Plan {
    std::vector<int> fn;             // index into kFns per call
    std::vector<int> first;          // offset of the call's first argument word
    std::vector<uint64_t> words;
};
static std::mutex g_plan_mu;
static std::vector<Plan*> g_plans;   // handle = index + 1

}  // namespace xl

using namespace xl;

// ---------------------------------------------------------------- stream plumbing the step needs besides kernels
extern "C" int xl_memset(void* dst, int value, int64_t bytes, void* stream) {
    XL_CHECK_ARG(dst != nullptr && bytes >= 0, XL_ERR_BAD_ARG, "xl_memset: bad args");
    if (bytes == 0) return XL_OK;
    hipError_t e = hipMemsetAsync(dst, value, (size_t)bytes, (hipStream_t)stream);
    XL_CHECK_ARG(e == hipSuccess, XL_ERR_HIP, "xl_memset: %s", hipGetErrorString(e));
    return XL_OK;
}

extern "C" int64_t xl_event_create(void) {
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
        set_error("xl_event_create: hipEventCreateWithFlags failed");
        return 0;
    }
    return (int64_t)reinterpret_cast<uintptr_t>(ev);
}

extern "C" int xl_event_destroy(int64_t event) {
    if (event != 0) (void)hipEventDestroy(reinterpret_cast<hipEvent_t>(static_cast<uintptr_t>(event)));
    return XL_OK;
}

extern "C" int xl_stream_fork(void* event, void* from_stream, void* to_stream) {
    XL_CHECK_ARG(event != nullptr, XL_ERR_BAD_ARG, "xl_stream_fork: null event");
    hipError_t e = hipEventRecord((hipEvent_t)event, (hipStream_t)from_stream);
    if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)to_stream, (hipEvent_t)event, 0);
    XL_CHECK_ARG(e == hipSuccess, XL_ERR_HIP, "xl_stream_fork: %s", hipGetErrorString(e));
    return XL_OK;
}

extern "C" int xl_event_record(void* event, void* stream) {
    XL_CHECK_ARG(event != nullptr, XL_ERR_BAD_ARG, "xl_event_record: null event");
    hipError_t e = hipEventRecord((hipEvent_t)event, (hipStream_t)stream);
    XL_CHECK_ARG(e == hipSuccess, XL_ERR_HIP, "xl_event_record: %s", hipGetErrorString(e));
    return XL_OK;
}

extern "C" int xl_stream_wait(void* event, void* stream) {
    XL_CHECK_ARG(event != nullptr, XL_ERR_BAD_ARG, "xl_stream_wait: null event");
    hipError_t e = hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0);
    XL_CHECK_ARG(e == hipSuccess, XL_ERR_HIP, "xl_stream_wait: %s", hipGetErrorString(e));
    return XL_OK;
}

// ---------------------------------------------------------------- plans
extern "C" int xl_plan_fn_id(const char* name) {
    for (int i = 0; i < kNumFns; ++i)
        if (strcmp(kFns[i].name, name) == 0) return i;
    set_error("xl_plan_fn_id: %s cannot be part of a launch plan", name ? name : "(null)");
    return XL_ERR_BAD_ARG;
}

extern "C" int xl_plan_fn_nargs(int fn_id) {
    XL_CHECK_ARG(fn_id >= 0 && fn_id < kNumFns, XL_ERR_BAD_ARG, "xl_plan_fn_nargs: bad id %d", fn_id);
    return kFns[fn_id].nargs;
}

extern "C" int64_t xl_plan_create(int n_calls, const int* fn_ids, const int* n_args, const uint64_t* words) {
    if (n_calls <= 0 || !fn_ids || !n_args || !words) { set_error("xl_plan_create: bad args"); return 0; }
    Plan* p = new Plan();
    int off = 0;
    for (int i = 0; i < n_calls; ++i) {
        if (fn_ids[i] < 0 || fn_ids[i] >= kNumFns || n_args[i] != kFns[fn_ids[i]].nargs) {
            set_error("xl_plan_create: call %d: function id %d with %d arguments", i, fn_ids[i], n_args[i]);
            delete p;
            return 0;
        }
        p->fn.push_back(fn_ids[i]);
        p->first.push_back(off);
        off += n_args[i];
    }
    p->words.assign(words, words + off);
    std::lock_guard<std::mutex> lk(g_plan_mu);
    g_plans.push_back(p);
    return (int64_t)g_plans.size();
}

extern "C" int xl_plan_run(int64_t plan) {
    Plan* p = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_plan_mu);
        if (plan >= 1 && plan <= (int64_t)g_plans.size()) p = g_plans[plan - 1];
    }
    XL_CHECK_ARG(p != nullptr, XL_ERR_BAD_ARG, "xl_plan_run: unknown plan %lld", (long long)plan);
    const size_t n = p->fn.size();
    // debug (XL_PLAN_TRACE=<file>, XL_PLAN_TRACE_RUN=<k>): host time of every call of the k-th replay in this process -- where the
    // runtime makes the enqueueing thread wait (tools/plan_host_trace.py)
    static const char* trace_file = getenv("XL_PLAN_TRACE");
    if (trace_file != nullptr) {
        static const int trace_run = getenv("XL_PLAN_TRACE_RUN") ? atoi(getenv("XL_PLAN_TRACE_RUN")) : 8;
        static int run = 0;
        if (++run == trace_run) {
            using clk = std::chrono::steady_clock;
            std::vector<double> t(n + 1);
            const auto t0 = clk::now();
            for (size_t i = 0; i < n; ++i) {
                t[i] = std::chrono::duration<double, std::micro>(clk::now() - t0).count();
                const int rc = kFns[p->fn[i]].call(p->words.data() + p->first[i]);
                if (rc < 0) return rc;
            }
            t[n] = std::chrono::duration<double, std::micro>(clk::now() - t0).count();
            if (FILE* f = fopen(trace_file, "w")) {
                for (size_t i = 0; i < n; ++i) {
                    const uint64_t* w = p->words.data() + p->first[i];
                    const uint64_t last = kFns[p->fn[i]].nargs > 0 ? w[kFns[p->fn[i]].nargs - 1] : 0;     // (the stream, for launches)
                    fprintf(f, "%zu %s %.1f %.1f %llx\n", i, kFns[p->fn[i]].name, t[i], t[i + 1] - t[i], (unsigned long long)last);
                }
                fclose(f);
            }
            return XL_OK;
        }
    }
    for (size_t i = 0; i < n; ++i) {
        const int rc = kFns[p->fn[i]].call(p->words.data() + p->first[i]);
        if (rc < 0) return rc;                  // xl_last_error() carries the failing call's message
    }
    return XL_OK;
}

extern "C" int xl_plan_destroy(int64_t plan) {
    std::lock_guard<std::mutex> lk(g_plan_mu);
    if (plan >= 1 && plan <= (int64_t)g_plans.size() && g_plans[plan - 1] != nullptr) {
        delete g_plans[plan - 1];
        g_plans[plan - 1] = nullptr;
    }
    return XL_OK;
}
