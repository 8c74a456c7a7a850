// Connect-Four on bitboards, one lane per game (gfx950).
//
// Board encoding: 7 columns x 7 bits (6 playable rows + one always-empty sentinel row), bit
// index = column*7 + height, height 0 = bottom row.  With this packing four-in-a-row tests
// are three shifts per direction: vertical 1, horizontal 7, diagonals 6 and 8.
#include "mz_common.h"

namespace {

__device__ __forceinline__ bool four_in_row(unsigned long long b) {
    unsigned long long m;
    m = b & (b >> 1);  if (m & (m >> 2)) return true;    // vertical
    m = b & (b >> 7);  if (m & (m >> 14)) return true;   // horizontal
    m = b & (b >> 6);  if (m & (m >> 12)) return true;   // diagonal
    m = b & (b >> 8);  if (m & (m >> 16)) return true;   // anti-diagonal
    return false;
}

constexpr unsigned long long kFullBoard = 0x0FDFBF7EFDFBFULL;  // 6 playable bits in each of 7 columns

__global__ void k_c4_reset(mz_c4_state s, const int8_t *__restrict__ mask) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= s.G) return;
    if (mask && !mask[g]) return;
    s.pieces[2 * (size_t)g] = 0ull;
    s.pieces[2 * (size_t)g + 1] = 0ull;
    s.to_move[g] = 0;
    s.done[g] = 0;
}

// One adapter-level step of game g: drop a piece, decide win / draw / illegal move, pass the
// turn, and report what the player who is now to move sees (reward, done).
__device__ __forceinline__ void c4_step(const mz_c4_state &s, int g, int col, float &reward, int8_t &done_out) {
    if (s.done[g]) {
        reward = 0.0f;
        done_out = 1;
        return;
    }
    const int mover = s.to_move[g];
    unsigned long long own = s.pieces[2 * (size_t)g + mover];
    const unsigned long long opp = s.pieces[2 * (size_t)g + 1 - mover];
    float r = 0.0f;
    int8_t done = 0;
    int8_t next_mover = (int8_t)(1 - mover);
    int height = 6;
    if (col >= 0 && col < 7) height = __popcll(((own | opp) >> (col * 7)) & 0x3Full);
    if (height >= 6) {
        // illegal move (full column or out of range): game over, mover -1, opponent 0, and
        // the cursor jumps to player_0, whose view is reported
        done = 1;
        next_mover = 0;
        r = (mover == 0) ? -1.0f : 0.0f;
    } else {
        own |= 1ull << (col * 7 + height);
        s.pieces[2 * (size_t)g + mover] = own;
        if (four_in_row(own)) {
            done = 1;
            r = -1.0f;  // seen by the player who just lost
        } else if ((own | opp) == kFullBoard) {
            done = 1;
        }
    }
    s.to_move[g] = next_mover;
    s.done[g] = done;
    reward = r;
    done_out = done;
}

__global__ void k_c4_step(mz_c4_state s, const int32_t *__restrict__ action, float *__restrict__ reward,
                          int8_t *__restrict__ done_out) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= s.G) return;
    float r;
    int8_t d;
    c4_step(s, g, action[g], r, d);
    reward[g] = r;
    done_out[g] = d;
}

// The end of a self-play move in one launch, one lane per game: the action pick of k_pick_action (first maximum of
// the policy, or the drawn action when the coin says so), the step above, the episode bookkeeping of k_episode_end
// and the restart of an ended game (k_c4_reset) -- the same results as those four kernels in a row.
__global__ void k_c4_move_end(mz_c4_state s, const float *__restrict__ policy, int A, const int8_t *__restrict__ explore,
                              const int32_t *__restrict__ random_action, int32_t *__restrict__ step_in_episode, int max_steps,
                              int32_t *__restrict__ action_out, int32_t *__restrict__ action_copy, float *__restrict__ reward,
                              int8_t *__restrict__ ended) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= s.G) return;
    int a;
    if (explore && explore[g]) {
        a = random_action[g];
    } else {
        const float *p = policy + (size_t)g * A;
        a = 0;
        for (int i = 1; i < A; ++i)
            if (p[i] > p[a]) a = i;
    }
    action_out[g] = a;
    if (action_copy) action_copy[g] = a;
    float r;
    int8_t d;
    c4_step(s, g, a, r, d);
    reward[g] = r;
    const int t = step_in_episode[g] + 1;
    const bool e = d != 0 || t >= max_steps;
    ended[g] = e ? 1 : 0;
    step_in_episode[g] = e ? 0 : t;
    if (e) {
        s.pieces[2 * (size_t)g] = 0ull;
        s.pieces[2 * (size_t)g + 1] = 0ull;
        s.to_move[g] = 0;
        s.done[g] = 0;
    }
}

// obs f32[G][2][6][7]: plane 0 = pieces of the player to move, plane 1 = the opponent's,
// row 0 = top of the board.  One lane per output element (coalesced stores).
__global__ void k_c4_observe(mz_c4_state s, float *__restrict__ obs, int8_t *__restrict__ to_play) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)s.G * 84;
    if (i >= total) return;
    const int g = (int)(i / 84), e = (int)(i % 84);
    const int plane = e / 42, cell = e % 42, row = cell / 7, col = cell % 7;
    const int mover = s.to_move[g];
    const unsigned long long b = s.pieces[2 * (size_t)g + (plane == 0 ? mover : 1 - mover)];
    obs[i] = ((b >> (col * 7 + (5 - row))) & 1ull) ? 1.0f : 0.0f;
    if (to_play && e == 0) to_play[g] = mover == 0 ? 1 : -1;
}

int check_state(const mz_c4_state *s) {
    if (!s || s->G <= 0 || !s->pieces || !s->to_move || !s->done)
        return mz::fail(MZ_ERR_ARG, "connect-four state has G <= 0 or a NULL array");
    return MZ_OK;
}

}  // namespace

extern "C" {

int mz_c4_reset(const mz_c4_state *state, const int8_t *mask, void *stream) {
    MZ_TRY(check_state(state));
    hipLaunchKernelGGL(k_c4_reset, dim3((state->G + 255) / 256), dim3(256), 0, (hipStream_t)stream, *state, mask);
    return mz::check_launch("k_c4_reset");
}

int mz_c4_step(const mz_c4_state *state, const int32_t *action, float *reward, int8_t *done, void *stream) {
    MZ_TRY(check_state(state));
    if (!action || !reward || !done) return mz::fail(MZ_ERR_ARG, "mz_c4_step: NULL argument");
    hipLaunchKernelGGL(k_c4_step, dim3((state->G + 255) / 256), dim3(256), 0, (hipStream_t)stream, *state, action,
                       reward, done);
    return mz::check_launch("k_c4_step");
}

int mz_c4_move_end(const mz_c4_state *state, const float *policy, int32_t A, const int8_t *explore, const int32_t *random_action,
                   int32_t *step_in_episode, int32_t max_steps, int32_t *action, int32_t *action_copy, float *reward,
                   int8_t *ended, void *stream) {
    MZ_TRY(check_state(state));
    if (!policy || A <= 0 || (explore && !random_action) || !step_in_episode || !action || !reward || !ended)
        return mz::fail(MZ_ERR_ARG, "mz_c4_move_end: NULL argument or non-positive action count");
    hipLaunchKernelGGL(k_c4_move_end, dim3((state->G + 255) / 256), dim3(256), 0, (hipStream_t)stream, *state, policy, A, explore,
                       random_action, step_in_episode, max_steps, action, action_copy, reward, ended);
    return mz::check_launch("k_c4_move_end");
}

int mz_c4_observe(const mz_c4_state *state, float *obs, int8_t *to_play, void *stream) {
    MZ_TRY(check_state(state));
    if (!obs) return mz::fail(MZ_ERR_ARG, "mz_c4_observe: obs is NULL");
    long long total = (long long)state->G * 84;
    hipLaunchKernelGGL(k_c4_observe, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *state,
                       obs, to_play);
    return mz::check_launch("k_c4_observe");
}

}  // extern "C"
