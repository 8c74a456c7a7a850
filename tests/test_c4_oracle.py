"""Hand-made known-answer boards for the Connect-Four rule restatement (CPU oracle).
Parity with PettingZoo's connect_four_v3 is unpinned: the package is neither vendored in
the reference nor installed, and the reference's tests fix only shapes."""
import numpy as np

from oracle.c4 import C4


def play(moves):
    g = C4()
    out = None
    for m in moves:
        out = g.step(m)
    return g, out


def test_vertical_win_reports_minus_one_to_next_player():
    g, (r, d) = play([0, 1, 0, 1, 0, 1, 0])
    assert (r, d) == (-1.0, True)
    assert g.to_play == -1  # player_1 is "to move" and sees the loss
    obs = g.observe()
    assert obs[1, 2:6, 0].sum() == 4 and obs[0, 3:6, 1].sum() == 3  # plane 0 = player to move


def test_horizontal_and_diagonals():
    assert play([0, 0, 1, 1, 2, 2, 3])[1] == (-1.0, True)
    # rising diagonal for player_0: (col,height) (0,0),(1,1),(2,2),(3,3)
    assert play([0, 1, 1, 2, 2, 3, 2, 3, 3, 6, 3])[1] == (-1.0, True)
    # falling diagonal for player_0: (3,0),(2,1),(1,2),(0,3)
    assert play([3, 2, 2, 1, 1, 0, 1, 0, 0, 6, 0])[1] == (-1.0, True)
    # player_1 wins too
    assert play([6, 0, 6, 0, 5, 0, 5, 0])[1] == (-1.0, True)


def test_no_win_midgame_and_observation_layout():
    g, (r, d) = play([3])
    assert (r, d) == (0.0, False) and g.to_play == -1
    obs = g.observe()
    assert obs[1, 5, 3] == 1.0 and obs.sum() == 1.0  # bottom row is row 5; mover's piece is now "opponent"


def test_draw_full_board():
    g = C4()
    # fill the columns pairwise with alternating offsets: no four in a row anywhere
    moves = [0, 1, 0, 1, 1, 0, 1, 0, 0, 1, 0, 1,
             2, 3, 2, 3, 3, 2, 3, 2, 2, 3, 2, 3,
             4, 5, 4, 5, 5, 4, 5, 4, 4, 5, 4, 5,
             6, 6, 6, 6, 6, 6]
    out = None
    for i, m in enumerate(moves):
        out = g.step(m)
        if i < len(moves) - 1:
            assert out == (0.0, False), (i, m, out)
    assert out == (0.0, True)
    assert (g.board != 0).all()


def test_illegal_move_full_column():
    g, (r, d) = play([0, 0, 0, 0, 0, 0, 0])  # 7th drop into column 0 by player_0
    assert (r, d) == (-1.0, True) and g.to_play == 1
    g, (r, d) = play([0, 0, 0, 0, 0, 0, 1, 0])  # player_1 makes the illegal move
    assert (r, d) == (0.0, True) and g.to_play == 1
    before = g.board.copy()
    assert g.step(3) == (0.0, True)  # finished games ignore further steps
    assert np.array_equal(before, g.board)


# Number of distinct Connect-Four positions on the 7x6 board after n plies when play stops at the
# first four-in-a-row (John Tromp's table, OEIS A212693).  Independent of PettingZoo: it pins move
# legality (column heights) and the win test (a position with a four-in-a-row is counted but not
# continued) against published numbers.
TROMP_POSITIONS = [1, 7, 49, 238, 1120, 4263, 16422, 54859]


def test_position_counts_match_published_table():
    import ctypes as C

    from oracle.c4 import C4, _State
    from oracle import lib
    step = lib().mzo_c4_step
    step.restype = C.c_float
    root = C4()
    frontier = {bytes(root.s.board): bytes(root.s)}
    counts = [1]
    for ply in range(1, len(TROMP_POSITIONS)):
        nxt = {}
        for raw in frontier.values():
            for a in range(7):
                s = _State.from_buffer_copy(raw)
                if s.board[a] != 0:  # top cell of the column occupied: not a legal move
                    continue
                d = C.c_int(0)
                step(C.byref(s), C.c_int(a), C.byref(d))
                nxt[bytes(s.board)] = (bytes(s), bool(d.value))
        counts.append(len(nxt))
        frontier = {k: raw for k, (raw, done) in nxt.items() if not done}
    assert counts == TROMP_POSITIONS
