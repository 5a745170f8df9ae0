// Stream ordering events for the engine's fork/join between its HIP streams (main stream: forward + data-gradient chain,
// side streams: weight gradients, histograms).  The reference has no counterpart (TensorFlow's executor orders its own ops:
// pix2pix_model.py:63-87 is one tf.function); this is the drop-in's own schedule.
//
// Why not torch.cuda.Event: its record carries a SYSTEM-scope release (L2 write-back + invalidate, so that a host reader would
// see the data).  These events only order kernels of one device -- every kernel's own dispatch packet already has agent-scope
// acquire/release -- so they are created with hipEventDisableSystemFence: r03 kernel traces show ~6 us of main-stream idle
// time after every record with the fence (17 per step).
#include "p2p_common.hpp"

extern "C" int p2p_event_create(void** ev_out) {
    P2P_REQUIRE(ev_out, "p2p_event_create: null pointer");
    hipEvent_t ev = nullptr;
    hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventDisableSystemFence);
    P2P_REQUIRE(e == hipSuccess, "p2p_event_create: %s", hipGetErrorString(e));
    *ev_out = (void*)ev;
    return 0;
}

extern "C" int p2p_event_destroy(void* ev) {
    if (!ev) return 0;
    hipError_t e = hipEventDestroy((hipEvent_t)ev);
    P2P_REQUIRE(e == hipSuccess, "p2p_event_destroy: %s", hipGetErrorString(e));
    return 0;
}

// Marks the work issued so far on `stream`.
extern "C" int p2p_event_record(void* ev, void* stream) {
    P2P_REQUIRE(ev, "p2p_event_record: null event");
    hipError_t e = hipEventRecord((hipEvent_t)ev, (hipStream_t)stream);
    P2P_REQUIRE(e == hipSuccess, "p2p_event_record: %s", hipGetErrorString(e));
    return 0;
}

// Work issued to `stream` after this call starts only when the event's latest record has completed.
extern "C" int p2p_stream_wait_event(void* stream, void* ev) {
    P2P_REQUIRE(ev, "p2p_stream_wait_event: null event");
    hipError_t e = hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0);
    P2P_REQUIRE(e == hipSuccess, "p2p_stream_wait_event: %s", hipGetErrorString(e));
    return 0;
}

// ---- the completion signal of a kernel as the fork event (P2P_LAUNCH_LAST, p2p_common.hpp) -----------------------------------------
static thread_local hipEvent_t g_stop_event = nullptr;

hipEvent_t p2p_take_stop_event() {
    hipEvent_t ev = g_stop_event;
    g_stop_event = nullptr;
    return ev;
}

// The next entry point of this thread whose last launch supports it (p2p_norm_act_bwd, p2p_act_bwd) signals `ev` when that kernel
// completes -- as p2p_event_record(ev, stream) right behind the call would, without a packet of its own on the stream.
extern "C" int p2p_arm_stop_event(void* ev) {
    P2P_REQUIRE(ev, "p2p_arm_stop_event: null event");
    g_stop_event = (hipEvent_t)ev;
    return 0;
}

// Behind that entry point: *was_pending = 1 if no launch took the event (the entry point does not support it, or took a path that
// does not): the caller records the event the ordinary way.  Leaves nothing armed.
extern "C" int p2p_disarm_stop_event(int* was_pending) {
    if (was_pending) *was_pending = g_stop_event != nullptr;
    g_stop_event = nullptr;
    return 0;
}
