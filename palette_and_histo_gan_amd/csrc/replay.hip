// Step replay: the host records the C-ABI calls of one train step as an array of p2p_replay_call (include/p2pgan.h) and
// re-issues the whole step with ONE call.  The reference runs train_step as one traced tf.function (pix2pix_model.py:62), i.e.
// its host pays one call per step; issued launch by launch from Python this build's step costs ~1.2 ms of interpreter time for
// ~145 calls, which is what the GPU needs for a whole batch-4 step and, on a slow host, more than the side stream can hide at
// batch 256 (BENCH_r04: the driver's box read 2.34 ms per step against 1.98 ms on a faster host).
//
// No code generation and no libffi: a thunk per entry point is instantiated from the entry point's own prototype (so a
// changed signature in p2pgan.h changes the thunk with it) and reads argument k from the 8-byte slot k of the record.
#include "p2p_common.hpp"
#include <string.h>
#include <utility>

namespace {

template <typename T>
inline T slot_as(unsigned long long v) {
    static_assert(sizeof(T) <= 8, "replayable entry points take scalars and pointers only");
    T t;
    memcpy(&t, &v, sizeof(T));      // little endian: ints and floats sit in the low bytes of the slot
    return t;
}

template <typename... A, size_t... I>
inline int call_slots(int (*fn)(A...), const unsigned long long* a, std::index_sequence<I...>) {
    return fn(slot_as<A>(a[I])...);
}

template <typename... A>
constexpr int arg_count(int (*)(A...)) { return (int)sizeof...(A); }

template <auto Fn>
struct Thunk {
    template <typename... A>
    static int go(int (*fn)(A...), const unsigned long long* a) { return call_slots(fn, a, std::index_sequence_for<A...>{}); }
    static int call(const unsigned long long* a) { return go(Fn, a); }
};

struct Entry {
    const char* name;
    int (*call)(const unsigned long long*);
    int nargs;
};

#define P2P_E(f) {#f, &Thunk<&f>::call, arg_count(&f)},
// every entry point of p2pgan.h that returns int (tests/test_host_cpu.py checks the list against the header)
constexpr Entry TABLE[] = {
    P2P_E(p2p_conv_direct) P2P_E(p2p_igemm) P2P_E(p2p_igemm_norm_act)
    P2P_E(p2p_igemm_edge) P2P_E(p2p_conv_strip) P2P_E(p2p_conv_fewin)
    P2P_E(p2p_conv_fewin_actbwd) P2P_E(p2p_conv_fewout) P2P_E(p2p_wgemm_edge)
    P2P_E(p2p_wgrad_small) P2P_E(p2p_view_colsum) P2P_E(p2p_act_bwd)
    P2P_E(p2p_weight_prep_pad) P2P_E(p2p_wgemm) P2P_E(p2p_norm_act_fwd)
    P2P_E(p2p_norm_act_fwd_tail) P2P_E(p2p_norm_act_bwd) P2P_E(p2p_colsum)
    P2P_E(p2p_colsum_batched) P2P_E(p2p_bce_logits) P2P_E(p2p_bce_logits_pad8)
    P2P_E(p2p_loss_partials_sum) P2P_E(p2p_tanh_l1_fwd) P2P_E(p2p_tanh_l1_fwd_pair)
    P2P_E(p2p_tanh_l1_bwd) P2P_E(p2p_tanh_l1_bwd_pad8) P2P_E(p2p_adam_flat)
    P2P_E(p2p_adam_tick) P2P_E(p2p_adam_flat_dev) P2P_E(p2p_counter_add)
    P2P_E(p2p_dropout_mask_dev) P2P_E(p2p_weight_prep) P2P_E(p2p_weight_prep_batched)
    P2P_E(p2p_adam_prep_batched) P2P_E(p2p_pack_input) P2P_E(p2p_pack_pair)
    P2P_E(p2p_pack_pair_idx) P2P_E(p2p_pack_input_multi) P2P_E(p2p_finish_losses)
    P2P_E(p2p_unpack) P2P_E(p2p_dropout_mask) P2P_E(p2p_rgbuv_hist_fwd)
    P2P_E(p2p_hist_normalize) P2P_E(p2p_rgbuv_hist_general) P2P_E(p2p_rgbuv_hist_fwd3) P2P_E(p2p_rgbuv_points)
    P2P_E(p2p_hellinger_fwd) P2P_E(p2p_hellinger_finish) P2P_E(p2p_rgbuv_hist_hellinger_bwd)
    P2P_E(p2p_rgbuv_hist_hellinger_bwd3) P2P_E(p2p_softmax_cce_argmax) P2P_E(p2p_argmax_lastdim)
    P2P_E(p2p_head_dgrad) P2P_E(p2p_head_softmax_cce) P2P_E(p2p_comm_unique_id)
    P2P_E(p2p_comm_init) P2P_E(p2p_comm_allreduce_sum) P2P_E(p2p_comm_destroy)
    P2P_E(p2p_event_create) P2P_E(p2p_event_destroy) P2P_E(p2p_event_record)
    P2P_E(p2p_stream_wait_event) P2P_E(p2p_arm_stop_event) P2P_E(p2p_disarm_stop_event) P2P_E(p2p_png_unfilter) P2P_E(p2p_sprites_rgba_batch)
    P2P_E(p2p_gather_rows_i32) P2P_E(p2p_palette_relabel_batch)
};
#undef P2P_E
constexpr int NFN = (int)(sizeof(TABLE) / sizeof(TABLE[0]));

constexpr bool table_fits() {
    for (int i = 0; i < NFN; ++i)
        if (TABLE[i].nargs > P2P_REPLAY_MAX_ARGS) return false;
    return true;
}
static_assert(table_fits(), "an entry point has more arguments than P2P_REPLAY_MAX_ARGS");

}  // namespace

extern "C" int p2p_replay_fn_index(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < NFN; ++i)
        if (strcmp(TABLE[i].name, name) == 0) return i;
    return -1;
}

extern "C" int p2p_replay_fn_nargs(int fn) { return (fn >= 0 && fn < NFN) ? TABLE[fn].nargs : -1; }

extern "C" int p2p_replay(const p2p_replay_call* calls, int n) {
    P2P_REQUIRE(calls || n == 0, "p2p_replay: null call list");
    for (int i = 0; i < n; ++i) {
        const p2p_replay_call& c = calls[i];
        P2P_REQUIRE(c.fn >= 0 && c.fn < NFN && c.nargs == TABLE[c.fn].nargs, "p2p_replay: call %d: bad entry %d / %d arguments", i,
                    c.fn, c.nargs);
        const unsigned long long* ap = c.a;
        unsigned long long a[P2P_REPLAY_MAX_ARGS];
        if (c.ind64 | c.ind32) {
            for (int k = 0; k < c.nargs; ++k) {
                unsigned long long v = c.a[k];
                if ((c.ind64 >> k) & 1u) v = *(const unsigned long long*)(uintptr_t)v;
                else if ((c.ind32 >> k) & 1u) v = *(const unsigned*)(uintptr_t)v;
                a[k] = v;
            }
            ap = a;
        }
        const int rc = TABLE[c.fn].call(ap);
        if (rc != 0) {
            char why[400];
            strncpy(why, p2p_last_error(), sizeof(why) - 1);
            why[sizeof(why) - 1] = 0;
            p2p_set_error("p2p_replay: call %d of %d (%s) failed: %s", i, n, TABLE[c.fn].name, why);
            return rc;
        }
    }
    return 0;
}
