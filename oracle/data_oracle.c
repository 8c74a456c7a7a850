/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement of the data-side pieces of the hot path:
 * unroll value targets and the observation-history tensor.  Loaded only by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * Parity status: PINNED by tests/golden/ vectors captured from the imported reference
 * (tools/make_goldens.py): z-targets from NeuralNetworkManager._compute_z_targets, history
 * tensors from FrameRingBuffer + make_history_tensor.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* src/training.py:223-242.  R and V hold w+1 entries (chunks k..k+w).  For each unroll
 * step k: z_k = sum_{i>=0} gamma^(i-1) * R[k+i]  +  gamma^(w-k) * V[w], accumulated left
 * to right in float64 starting from 0.0 (note the exponent i-1: the first reward is
 * divided by gamma). */
void mzo_z_targets(const double *R, const double *V, int32_t w, double gamma, double *z) {
    for (int k = 0; k <= w; ++k) {
        volatile double acc = 0.0;
        for (int i = 0; k + i <= w; ++i) {
            volatile double term = pow(gamma, (double)(i - 1)) * R[k + i];
            acc = acc + term;
        }
        volatile double boot = pow(gamma, (double)(w - k)) * V[w];
        acc = acc + boot;
        z[k] = acc;
    }
}

/* src/ring_buffer.py:11-42.  A K-slot ring: fill() replicates one frame into every empty
 * slot and leaves the write cursor alone; add() overwrites slot `cursor` and advances it.
 * `slot_of_step[t]` receives the slot that step t (t = 0 is the fill) wrote, so callers
 * can build the slot-ordered frame list the history tensor uses. */
void mzo_ring_slots(int32_t K, int32_t T, int32_t *slot_of_step) {
    int32_t cursor = 0;
    slot_of_step[0] = -1; /* fill: every slot */
    for (int t = 1; t < T; ++t) {
        slot_of_step[t] = cursor;
        cursor = (cursor + 1) % K;
    }
}

/* src/ring_buffer.py:44-83.  frames: [K][C][H][W] in SLOT order, actions: [K] in slot
 * order.  out: [C*K + K][H][W] = frames concatenated on the channel axis followed by K
 * planes each filled with the raw action id of its slot. */
void mzo_history(const float *frames, const int32_t *actions, int32_t K, int32_t Cc, int32_t H,
                 int32_t W, float *out) {
    size_t plane = (size_t)H * W;
    memcpy(out, frames, sizeof(float) * K * Cc * plane);
    float *act = out + (size_t)K * Cc * plane;
    for (int k = 0; k < K; ++k)
        for (size_t i = 0; i < plane; ++i) act[k * plane + i] = (float)actions[k];
}
