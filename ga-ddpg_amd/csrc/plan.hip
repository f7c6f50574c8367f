// Step replay (gaddpg.h section H): a recorded list of C-ABI launches, copies, clears and stream forks / joins, executed by ONE
// call instead of one foreign call per launch from the host language.  The update step of reference core/ddpg.py:146-185 is
// ~240 launches on four streams; the Python host used to walk them one ctypes call at a time (1.3 - 2 ms of interpreter time per
// 2.5 ms step, and under the GIL: a prefetch thread stalls the walk).  Here the walk is a C loop; the foreign call releases the GIL.
//
// Host-only code (compiled by hipcc for the HIP runtime headers; there is no kernel in this file).  Every entry point that a plan
// can call is registered below with a typed thunk generated from its real signature: the argument words of an item are checked
// against that signature when the item is ADDED (count and kind), so a replay cannot mis-call the ABI.
#include "common.hpp"

#include <string.h>

#include <string>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

namespace {

typedef int (*thunk_fn)(const uint64_t* w, void* stream);

template <typename T> struct arg_kind {
    static constexpr int value = std::is_pointer<T>::value ? GAD_ARG_I64
                                 : std::is_same<T, float>::value ? GAD_ARG_F32
                                 : std::is_same<T, double>::value ? GAD_ARG_F64
                                 : (sizeof(T) == 8 ? GAD_ARG_I64 : GAD_ARG_I32);
};

template <typename T> inline T word_to(uint64_t w) {
    if constexpr (std::is_pointer<T>::value) {
        return reinterpret_cast<T>(static_cast<uintptr_t>(w));
    } else if constexpr (std::is_same<T, float>::value) {
        uint32_t b = (uint32_t)w;
        float f;
        memcpy(&f, &b, 4);
        return f;
    } else if constexpr (std::is_same<T, double>::value) {
        double d;
        memcpy(&d, &w, 8);
        return d;
    } else {
        return (T)(int64_t)w;
    }
}

template <auto F> struct entry;
template <typename... A, int (*F)(A...)> struct entry<F> {
    static constexpr int N = (int)sizeof...(A) - 1;          // the trailing argument is the stream
    using tup = std::tuple<A...>;
    template <size_t... I> static int call(std::index_sequence<I...>, const uint64_t* w, void* stream) {
        return F(word_to<std::tuple_element_t<I, tup>>(w[I])..., stream);
    }
    static int run(const uint64_t* w, void* stream) { return call(std::make_index_sequence<(size_t)N>{}, w, stream); }
    template <size_t... I> static void kinds(std::index_sequence<I...>, uint8_t* k) {
        ((k[I] = (uint8_t)arg_kind<std::tuple_element_t<I, tup>>::value), ...);
    }
    static void fill_kinds(uint8_t* k) { kinds(std::make_index_sequence<(size_t)N>{}, k); }
};

struct reg_entry {
    const char* name;
    thunk_fn fn;
    int n;
    uint8_t kinds[32];
};

#define GAD_PLAN_ENTRY(f)                                        \
    {                                                            \
        reg_entry e;                                             \
        e.name = #f;                                             \
        e.fn = &entry<&f>::run;                                  \
        e.n = entry<&f>::N;                                      \
        static_assert(entry<&f>::N <= 32, "too many arguments"); \
        entry<&f>::fill_kinds(e.kinds);                          \
        v.push_back(e);                                          \
    }

const std::vector<reg_entry>& registry() {
    static const std::vector<reg_entry> r = [] {
        std::vector<reg_entry> v;
        GAD_PLAN_ENTRY(gad_grid_rows_hint)
        GAD_PLAN_ENTRY(gad_furthest_point_sampling)
        GAD_PLAN_ENTRY(gad_gather_points)
        GAD_PLAN_ENTRY(gad_gather_points_grad)
        GAD_PLAN_ENTRY(gad_ball_query)
        GAD_PLAN_ENTRY(gad_group_points)
        GAD_PLAN_ENTRY(gad_group_points_grad)
        GAD_PLAN_ENTRY(gad_query_and_group)
        GAD_PLAN_ENTRY(gad_prep_points)
        GAD_PLAN_ENTRY(gad_rows_from_ball_query)
        GAD_PLAN_ENTRY(gad_rows_group_all)
        GAD_PLAN_ENTRY(gad_gemm_fwd)
        GAD_PLAN_ENTRY(gad_bn_finalize)
        GAD_PLAN_ENTRY(gad_bn_running_update)
        GAD_PLAN_ENTRY(gad_bn_eval_affine)
        GAD_PLAN_ENTRY(gad_segment_pool)
        GAD_PLAN_ENTRY(gad_pool_finalize)
        GAD_PLAN_ENTRY(gad_affine_act)
        GAD_PLAN_ENTRY(gad_transpose_batched)
        GAD_PLAN_ENTRY(gad_pool_bwd_stats)
        GAD_PLAN_ENTRY(gad_bn_bwd_coef)
        GAD_PLAN_ENTRY(gad_gemm_dx)
        GAD_PLAN_ENTRY(gad_gemm_dw)
        GAD_PLAN_ENTRY(gad_gemm_bwd)
        GAD_PLAN_ENTRY(gad_gemm_dw_reduce)
        GAD_PLAN_ENTRY(gad_critic_loss)
        GAD_PLAN_ENTRY(gad_policy_outputs)
        GAD_PLAN_ENTRY(gad_actor_loss)
        GAD_PLAN_ENTRY(gad_policy_sample)
        GAD_PLAN_ENTRY(gad_actor_critic_loss)
        GAD_PLAN_ENTRY(gad_mask_counts)
        GAD_PLAN_ENTRY(gad_target_noise)
        GAD_PLAN_ENTRY(gad_grad_from_arena)
        GAD_PLAN_ENTRY(gad_grad_from_arena_sumsq)
        GAD_PLAN_ENTRY(gad_sumsq)
        GAD_PLAN_ENTRY(gad_absmax_segments)
        GAD_PLAN_ENTRY(gad_adam_step)
        GAD_PLAN_ENTRY(gad_polyak)
        GAD_PLAN_ENTRY(gad_optim_jobs)
        GAD_PLAN_ENTRY(gad_pack_params)
        GAD_PLAN_ENTRY(gad_split_weights)
        GAD_PLAN_ENTRY(gad_replay_gather)
        GAD_PLAN_ENTRY(gad_zero_buffers)
        GAD_PLAN_ENTRY(gad_copy_buffers)
        return v;
    }();
    return r;
}

enum item_kind { IT_CALL = 0, IT_WAIT, IT_RECORD, IT_WAIT_EVENT, IT_MEMSET, IT_MEMCPY };

struct item {
    int kind;
    int lane;              // the stream the item is enqueued on (IT_WAIT: the waiting lane)
    int lane2;             // IT_WAIT: the signalling lane
    int first, n;          // argument words of the item in plan::words
    const reg_entry* e;    // IT_CALL
    hipEvent_t ev;         // IT_WAIT: plan-owned event
    void* timing;          // IT_CALL: timing slot armed for this launch on the next run (consumed), or NULL
};

}        // namespace

struct gad_plan {
    std::vector<item> items;
    std::vector<uint64_t> words;
    std::vector<uint8_t> kinds;
    int max_lane = 0;
};

extern "C" int gad_plan_create(gad_plan** out) {
    GAD_REQUIRE(out, GAD_ERR_NULL, "plan_create: NULL output");
    *out = new gad_plan();
    return GAD_OK;
}

extern "C" int gad_plan_destroy(gad_plan* p) {
    if (!p) return GAD_OK;
    for (auto& it : p->items)
        if (it.kind == IT_WAIT && it.ev) (void)hipEventDestroy(it.ev);
    delete p;
    return GAD_OK;
}

extern "C" int gad_plan_size(const gad_plan* p) { return p ? (int)p->items.size() : GAD_ERR_NULL; }

static int plan_push(gad_plan* p, item it, const uint64_t* w, const uint8_t* k, int n) {
    it.first = (int)p->words.size();
    it.n = n;
    for (int i = 0; i < n; ++i) {
        p->words.push_back(w[i]);
        p->kinds.push_back(k ? k[i] : (uint8_t)GAD_ARG_I64);
    }
    if (it.lane > p->max_lane) p->max_lane = it.lane;
    if (it.kind == IT_WAIT && it.lane2 > p->max_lane) p->max_lane = it.lane2;
    p->items.push_back(it);
    return (int)p->items.size() - 1;
}

extern "C" int gad_plan_add_call(gad_plan* p, const char* entry_name, const uint64_t* words, const uint8_t* kinds, int n_words,
                                 int lane) {
    GAD_REQUIRE(p && entry_name, GAD_ERR_NULL, "plan_add_call: NULL plan / entry name");
    GAD_REQUIRE(lane >= 0 && lane < GAD_PLAN_MAX_LANES, GAD_ERR_SHAPE, "plan_add_call: lane %d outside 0..%d", lane,
                GAD_PLAN_MAX_LANES - 1);
    const reg_entry* e = nullptr;
    for (const auto& r : registry())
        if (strcmp(r.name, entry_name) == 0) { e = &r; break; }
    GAD_REQUIRE(e, GAD_ERR_UNSUPPORTED, "plan_add_call: '%s' is not an entry point a plan can replay", entry_name);
    GAD_REQUIRE(n_words == e->n, GAD_ERR_SHAPE, "plan_add_call: %s takes %d arguments before the stream, got %d", entry_name,
                e->n, n_words);
    GAD_REQUIRE(n_words == 0 || (words && kinds), GAD_ERR_NULL, "plan_add_call: NULL argument words / kinds");
    for (int i = 0; i < n_words; ++i)
        GAD_REQUIRE(kinds[i] == e->kinds[i], GAD_ERR_SHAPE, "plan_add_call: %s argument %d is of kind %d, the caller packed kind %d",
                    entry_name, i, (int)e->kinds[i], (int)kinds[i]);
    item it{};
    it.kind = IT_CALL;
    it.lane = lane;
    it.e = e;
    return plan_push(p, it, words, kinds, n_words);
}

extern "C" int gad_plan_add_wait(gad_plan* p, int waiter_lane, int signal_lane) {
    GAD_REQUIRE(p, GAD_ERR_NULL, "plan_add_wait: NULL plan");
    GAD_REQUIRE(waiter_lane >= 0 && waiter_lane < GAD_PLAN_MAX_LANES && signal_lane >= 0 && signal_lane < GAD_PLAN_MAX_LANES,
                GAD_ERR_SHAPE, "plan_add_wait: lanes %d / %d outside 0..%d", waiter_lane, signal_lane, GAD_PLAN_MAX_LANES - 1);
    item it{};
    it.kind = IT_WAIT;
    it.lane = waiter_lane;
    it.lane2 = signal_lane;
    if (hipEventCreateWithFlags(&it.ev, hipEventDisableTiming) != hipSuccess) {
        gad_set_error("plan_add_wait: hipEventCreateWithFlags failed");
        return GAD_ERR_LAUNCH;
    }
    return plan_push(p, it, nullptr, nullptr, 0);
}

static int add_simple(gad_plan* p, int kind, int lane, const uint64_t* w, int n, const char* what) {
    GAD_REQUIRE(p, GAD_ERR_NULL, "%s: NULL plan", what);
    GAD_REQUIRE(lane >= 0 && lane < GAD_PLAN_MAX_LANES, GAD_ERR_SHAPE, "%s: lane %d outside 0..%d", what, lane, GAD_PLAN_MAX_LANES - 1);
    item it{};
    it.kind = kind;
    it.lane = lane;
    return plan_push(p, it, w, nullptr, n);
}

extern "C" int gad_plan_add_record(gad_plan* p, int lane, void* event) {
    const uint64_t w[1] = {(uint64_t)(uintptr_t)event};
    return add_simple(p, IT_RECORD, lane, w, 1, "plan_add_record");
}

extern "C" int gad_plan_add_wait_event(gad_plan* p, int lane, void* event) {
    const uint64_t w[1] = {(uint64_t)(uintptr_t)event};
    return add_simple(p, IT_WAIT_EVENT, lane, w, 1, "plan_add_wait_event");
}

extern "C" int gad_plan_add_memset(gad_plan* p, void* dst, long long bytes, int lane) {
    GAD_REQUIRE(bytes >= 0, GAD_ERR_SHAPE, "plan_add_memset: negative size");
    const uint64_t w[2] = {(uint64_t)(uintptr_t)dst, (uint64_t)bytes};
    return add_simple(p, IT_MEMSET, lane, w, 2, "plan_add_memset");
}

extern "C" int gad_plan_add_memcpy(gad_plan* p, void* dst, const void* src, long long bytes, int lane) {
    GAD_REQUIRE(bytes >= 0, GAD_ERR_SHAPE, "plan_add_memcpy: negative size");
    const uint64_t w[3] = {(uint64_t)(uintptr_t)dst, (uint64_t)(uintptr_t)src, (uint64_t)bytes};
    return add_simple(p, IT_MEMCPY, lane, w, 3, "plan_add_memcpy");
}

extern "C" int gad_plan_patch(gad_plan* p, int item_index, int word, uint64_t value) {
    GAD_REQUIRE(p, GAD_ERR_NULL, "plan_patch: NULL plan");
    GAD_REQUIRE(item_index >= 0 && item_index < (int)p->items.size(), GAD_ERR_SHAPE, "plan_patch: item %d of %d", item_index,
                (int)p->items.size());
    const item& it = p->items[item_index];
    GAD_REQUIRE(word >= 0 && word < it.n, GAD_ERR_SHAPE, "plan_patch: word %d of an item with %d", word, it.n);
    p->words[it.first + word] = value;
    return GAD_OK;
}

extern "C" int gad_plan_arm_timing(gad_plan* p, int item_index, void* slot) {
    GAD_REQUIRE(p, GAD_ERR_NULL, "plan_arm_timing: NULL plan");
    GAD_REQUIRE(item_index >= 0 && item_index < (int)p->items.size() && p->items[item_index].kind == IT_CALL, GAD_ERR_SHAPE,
                "plan_arm_timing: item %d is not a launch", item_index);
    GAD_REQUIRE((reinterpret_cast<size_t>(slot) & 7) == 0, GAD_ERR_SHAPE, "plan_arm_timing: the slot must be 8-byte aligned");
    p->items[item_index].timing = slot;
    return GAD_OK;
}

extern "C" int gad_plan_run(gad_plan* p, void* const* streams, int n_streams, int first, int count) {
    GAD_REQUIRE(p && streams, GAD_ERR_NULL, "plan_run: NULL plan / stream table");
    const int n_items = (int)p->items.size();
    if (count < 0) count = n_items - first;
    GAD_REQUIRE(first >= 0 && count >= 0 && first + count <= n_items, GAD_ERR_SHAPE, "plan_run: items [%d, %d) of %d", first,
                first + count, n_items);
    GAD_REQUIRE(n_streams > p->max_lane, GAD_ERR_SHAPE, "plan_run: the plan uses lane %d, the table holds %d streams", p->max_lane,
                n_streams);
    const uint64_t* W = p->words.data();
    for (int i = first; i < first + count; ++i) {
        item& it = p->items[i];
        hipStream_t s = static_cast<hipStream_t>(streams[it.lane]);
        const uint64_t* w = W + it.first;
        hipError_t e = hipSuccess;
        switch (it.kind) {
            case IT_CALL: {
                if (it.timing) {
                    gad_timing_slot(it.timing);
                    it.timing = nullptr;
                }
                const int rc = it.e->fn(w, (void*)s);
                if (rc != GAD_OK) {
                    // (the entry point's own message is in the thread-local slot: keep it, prefix the position)
                    std::string msg = gad_last_error();
                    gad_set_error("plan item %d (%s, lane %d): %s", i, it.e->name, it.lane, msg.c_str());
                    return rc;
                }
                break;
            }
            case IT_WAIT:
                e = hipEventRecord(it.ev, static_cast<hipStream_t>(streams[it.lane2]));
                if (e == hipSuccess && streams[it.lane2] != streams[it.lane]) e = hipStreamWaitEvent(s, it.ev, 0);
                break;
            case IT_RECORD:
                if (w[0]) e = hipEventRecord(reinterpret_cast<hipEvent_t>((uintptr_t)w[0]), s);
                break;
            case IT_WAIT_EVENT:
                if (w[0]) e = hipStreamWaitEvent(s, reinterpret_cast<hipEvent_t>((uintptr_t)w[0]), 0);
                break;
            case IT_MEMSET:
                if (w[0] && w[1]) e = hipMemsetAsync(reinterpret_cast<void*>((uintptr_t)w[0]), 0, (size_t)w[1], s);
                break;
            case IT_MEMCPY:
                if (w[0] && w[1] && w[2])
                    e = hipMemcpyAsync(reinterpret_cast<void*>((uintptr_t)w[0]), reinterpret_cast<const void*>((uintptr_t)w[1]),
                                       (size_t)w[2], hipMemcpyDefault, s);
                break;
        }
        if (e != hipSuccess) {
            gad_set_error("plan item %d (kind %d, lane %d): %s", i, it.kind, it.lane, hipGetErrorString(e));
            return GAD_ERR_LAUNCH;
        }
    }
    return GAD_OK;
}

extern "C" int gad_plan_entry_count(void) { return (int)registry().size(); }
extern "C" const char* gad_plan_entry_name(int i) {
    return (i >= 0 && i < (int)registry().size()) ? registry()[i].name : nullptr;
}
