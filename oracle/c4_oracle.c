/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement of the Connect-Four rules the reference
 * gets from PettingZoo (pettingzoo==1.24.3, classic/connect_four_v3; the package is
 * not vendored in the reference and not installed here) as seen through the
 * reference's adapter src/environments/connect_four.py:18-72.
 *
 * Parity status: no reference test or fixture exercises the rules
 * (tests/search/test_mcts.py:22 and tests/environments pin only the observation shape
 * (2,6,7) and the 7 actions).  Move legality and the win test are pinned independently
 * by the published number of positions after n plies (John Tromp's table, OEIS A212693;
 * tests/test_c4_oracle.py).  UNPINNED: the reward / termination convention of the
 * PettingZoo wrappers restated below; its known-answer boards are hand-made.
 *
 * Board: 6 rows x 7 columns, flat cell index row*7+col with row 0 at the TOP.
 * Player 0 (`player_0`, to_play +1) moves first.  A move drops into the lowest empty
 * cell of its column.  The mover wins with four in a row (horizontal, vertical, both
 * diagonals); a full board without a winner is a draw.  Dropping into a full column
 * is an illegal move: the board is left unchanged, the game terminates, the mover is
 * assigned -1 and the opponent 0 (PettingZoo's TerminateIllegalWrapper with
 * illegal_reward=-1), and the agent cursor jumps to the first terminated agent, which
 * is always player_0 -- so the adapter then reports player_0's view: reward -1 when
 * player_0 made the illegal move, 0 when player_1 did.
 *
 * The adapter's step() returns what the NEXT agent sees through env.last()
 * (connect_four.py:44-51): the observation has plane 0 = pieces of the player now to
 * move, plane 1 = the opponent's; reward is that player's reward (a winning move
 * therefore yields -1); done is that player's termination flag.
 */
#include <stdint.h>

typedef struct {
    int8_t board[42]; /* 0 empty, 1 = player_0 piece, 2 = player_1 piece */
    int8_t to_move;   /* 0 or 1 */
    int8_t done;
} mzo_c4;

void mzo_c4_reset(mzo_c4 *s) {
    for (int i = 0; i < 42; ++i) s->board[i] = 0;
    s->to_move = 0;
    s->done = 0;
}

static int c4_wins(const int8_t *b, int piece) {
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 4; ++c)
            if (b[r * 7 + c] == piece && b[r * 7 + c + 1] == piece && b[r * 7 + c + 2] == piece &&
                b[r * 7 + c + 3] == piece)
                return 1;
    for (int c = 0; c < 7; ++c)
        for (int r = 0; r < 3; ++r)
            if (b[r * 7 + c] == piece && b[(r + 1) * 7 + c] == piece && b[(r + 2) * 7 + c] == piece &&
                b[(r + 3) * 7 + c] == piece)
                return 1;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c)
            if (b[r * 7 + c] == piece && b[(r + 1) * 7 + c + 1] == piece &&
                b[(r + 2) * 7 + c + 2] == piece && b[(r + 3) * 7 + c + 3] == piece)
                return 1;
    for (int r = 3; r < 6; ++r)
        for (int c = 0; c < 4; ++c)
            if (b[r * 7 + c] == piece && b[(r - 1) * 7 + c + 1] == piece &&
                b[(r - 2) * 7 + c + 2] == piece && b[(r - 3) * 7 + c + 3] == piece)
                return 1;
    return 0;
}

/* One adapter-level step.  Returns the reward seen by the player who is to move after
 * the step; *done_out is that player's termination flag.  Stepping a finished game is
 * a no-op that returns 0 with done set. */
float mzo_c4_step(mzo_c4 *s, int action, int *done_out) {
    if (s->done) { *done_out = 1; return 0.0f; }
    int mover = s->to_move, piece = mover + 1;
    float next_reward = 0.0f;
    int row = -1;
    for (int r = 5; r >= 0; --r)
        if (s->board[r * 7 + action] == 0) { row = r; break; }
    if (row < 0) {
        /* illegal move: mover -1, opponent 0, everyone terminates, cursor -> player_0 */
        s->done = 1;
        s->to_move = 0;
        *done_out = 1;
        return mover == 0 ? -1.0f : 0.0f;
    } else {
        s->board[row * 7 + action] = (int8_t)piece;
        if (c4_wins(s->board, piece)) {
            s->done = 1;
            next_reward = -1.0f;
        } else {
            int full = 1;
            for (int c = 0; c < 7; ++c)
                if (s->board[c] == 0) full = 0;
            if (full) s->done = 1;
        }
    }
    s->to_move = (int8_t)(1 - mover);
    *done_out = s->done;
    return next_reward;
}

/* Observation of the player to move: obs[0][r][c] = own pieces, obs[1][r][c] = the
 * opponent's, float32 (connect_four.py:53-57). */
void mzo_c4_observe(const mzo_c4 *s, float *obs) {
    int own = s->to_move + 1, opp = 2 - s->to_move;
    for (int i = 0; i < 42; ++i) {
        obs[i] = (s->board[i] == own) ? 1.0f : 0.0f;
        obs[42 + i] = (s->board[i] == opp) ? 1.0f : 0.0f;
    }
}

int mzo_c4_to_play(const mzo_c4 *s) { return s->to_move == 0 ? 1 : -1; }
