"""The launch-width controller's policy (csrc/width_policy.h) over synthetic measurement windows: CPU only, through
mrt_debug_width_policy.  The reference has ONE schedule -- one draw per State::redraw, one frame after the other
(lib.rs:241-307) -- so nothing here has a counterpart there; what is pinned is that the state machine this library puts in
its place decides what DESIGN.md 4 says it decides."""
import pytest

# (n_tiles, n_waves, max_slots, spp, n_members, counter)
C3 = (240 * 135, 5120, 16, 512, 488, 0)             # 1920x1080: 32,400 tiles for 5,120 waves
C2 = (150 * 85, 5120, 16, 64, 488, 0)               # 12,750 tiles: fewer than four per wave
C1 = (50 * 29, 5120, 16, 16, 4, 0)                  # 1,450 tiles: fewer than waves
C5 = (240 * 135, 5120, 16, 4096, 10004, 0)
C5_EIGHTH = (240 * 17, 5120, 16, 4096, 10004, 0)    # 4,080 tiles: pixel-starved, long chains
INTERACTIVE = (240 * 135, 5120, 16, 1, 488, 0)

DIV, MULT, PREV_DIV, PREV_MULT, LOW, SETTLED, PREV_RATE = range(7)


def slots(w, n):
    return w[:2] + (n,) + w[3:]


def start(mrt, w):
    return mrt.width_policy(0, w, None)


def step(mrt, w, s, util, rate):
    return mrt.width_policy(1, w, s, util, rate)


def test_what_is_known_up_front(mrt):
    assert start(mrt, C3)[:2] == [2, 2]                  # tiles to spare: a half, twice the launches the chip holds
    assert start(mrt, (480 * 34, 5120, 16, 1024, 488, 0))[:2] == [2, 2]     # C4's 1/8 share: 3.2 tiles per wave
    assert start(mrt, C5)[:2] == [4, 2]                  # ... a quarter for large scenes (their pixels' chains differ 10 x)
    assert start(mrt, slots(C5, 4))[:2] == [2, 2]
    assert start(mrt, C2)[:2] == [1, 1]
    assert start(mrt, C1)[:2] == [1, 1]
    assert start(mrt, C5_EIGHTH)[:2] == [8, 2]           # pixel-starved share of long chains: an eighth, sixteen in flight
    assert start(mrt, INTERACTIVE)[:2] == [4, 1]         # chains of a few bounces: a quarter (and the trials go on)
    assert start(mrt, C5_EIGHTH[:5] + (1,))[:2] == [1, 1]    # counter mode splits pixels into blocks: never starved
    for s in (start(mrt, w) for w in (C3, C1, C5, C5_EIGHTH, INTERACTIVE)):
        assert s[PREV_DIV] == 0 and s[SETTLED] == 0 and s[LOW] == 0


def test_a_process_with_fewer_hardware_queues_starts_within_them(mrt):
    """What probe_stream_concurrency found caps the frames in flight: HIP's default of 4 hardware queues -> 4."""
    assert start(mrt, slots(C5_EIGHTH, 8))[:2] == [8, 1]
    assert start(mrt, slots(C5_EIGHTH, 4))[:2] == [4, 1]
    assert start(mrt, slots(C5_EIGHTH, 2))[:2] == [2, 1]
    assert start(mrt, slots(C3, 4))[:2] == [2, 2]
    assert start(mrt, slots(C3, 2))[:2] == [2, 1]


def test_high_utilisation_settles_at_once(mrt):
    s = step(mrt, C3, start(mrt, C3), 0.986, 12.0)
    assert s[:2] == [2, 2] and s[SETTLED] == 1 and s[PREV_DIV] == 0
    # ... also right after a trial that paid (no exploration beyond what low utilisation asks for)
    s = step(mrt, C2, start(mrt, C2), 0.80, 200.0)
    s = step(mrt, C2, s, 0.98, 236.0)
    assert s[:2] == [4, 2] and s[SETTLED] == 1


def test_one_window_between_the_thresholds_only_asks_for_a_second(mrt):
    s = step(mrt, C3, start(mrt, C3), 0.93, 12.0)
    assert s[:2] == [2, 2] and s[SETTLED] == 0 and s[PREV_DIV] == 0 and s[LOW] == 1
    # the second window is high: nothing is tried
    t = step(mrt, C3, s, 0.97, 12.0)
    assert t[:2] == [2, 2] and t[SETTLED] == 1
    # the second window is low again: a quarter width (again with twice the launches that fit) is tried, the rate remembered
    t = step(mrt, C3, s, 0.94, 12.5)
    assert t[:2] == [4, 2] and t[SETTLED] == 0 and t[PREV_DIV] == 2 and t[PREV_MULT] == 2 and t[LOW] == 0
    assert t[PREV_RATE] == pytest.approx(12.5)


def test_below_the_lower_threshold_one_window_is_enough(mrt):
    s = step(mrt, C2, start(mrt, C2), 0.80, 200.0)
    assert s[:2] == [4, 2] and s[PREV_DIV] == 1 and s[PREV_MULT] == 1 and s[SETTLED] == 0


def test_a_trial_that_pays_is_kept_and_the_next_one_starts(mrt):
    s = step(mrt, C2, start(mrt, C2), 0.80, 200.0)          # trial: full width, 2 in flight -> a quarter, 8 in flight
    s = step(mrt, C2, s, 0.93, 236.0)                       # + 18 %: kept; 0.93 asks for a second window before the next trial
    assert s[:2] == [4, 2] and s[PREV_DIV] == 0 and s[SETTLED] == 0 and s[LOW] == 1
    s = step(mrt, C2, s, 0.93, 236.0)                       # second low window: C2 has 12,750 tiles, an eighth still leaves 2 per wave
    assert s[:2] == [8, 2] and s[PREV_DIV] == 4 and s[PREV_MULT] == 2 and s[PREV_RATE] == pytest.approx(236.0)
    s = step(mrt, C2, s, 0.99, 237.0)                       # + 0.4 %: not kept
    assert s[:2] == [4, 2] and s[SETTLED] == 1 and s[PREV_DIV] == 0


def test_a_trial_that_does_not_pay_is_reverted_and_the_setting_stays(mrt):
    s = step(mrt, C3, start(mrt, C3), 0.85, 12.0)           # trial: a half -> a quarter
    assert s[:2] == [4, 2] and s[PREV_DIV] == 2
    s = step(mrt, C3, s, 0.99, 12.2)                        # + 1.7 % < 3 %
    assert s[:2] == [2, 2] and s[SETTLED] == 1 and s[PREV_DIV] == 0
    # settled: further windows change nothing
    assert step(mrt, C3, s, 0.50, 1.0)[:2] == [2, 2]


def test_no_room_to_narrow_adds_frames_in_flight_instead(mrt):
    # C1: 1,450 tiles for 5,120 waves -- a quarter of the waves would leave less than two tiles per wave
    s = step(mrt, C1, start(mrt, C1), 0.60, 2000.0)
    assert s[:2] == [1, 2] and s[PREV_DIV] == 1 and s[PREV_MULT] == 1
    s = step(mrt, C1, s, 0.60, 3500.0)                      # kept; still low: four times the frames
    assert s[:2] == [1, 4] and s[PREV_MULT] == 2
    s = step(mrt, C1, s, 0.60, 6000.0)                      # kept: eight times
    assert s[:2] == [1, 8] and s[PREV_MULT] == 4
    s = step(mrt, C1, s, 0.60, 7000.0)                      # kept; 16 frames in flight is the most there is
    assert s[:2] == [1, 8] and s[SETTLED] == 1
    eight = slots(C1, 8)
    s = step(mrt, eight, [1, 4, 0, 1, 0, 0, 0.0], 0.60, 6000.0)
    assert s[:2] == [1, 4] and s[SETTLED] == 1


def test_the_narrowest_launch_is_an_eighth(mrt):
    s = step(mrt, INTERACTIVE, start(mrt, INTERACTIVE), 0.78, 3000.0)     # a quarter -> an eighth, sixteen in flight
    assert s[:2] == [8, 2] and s[PREV_DIV] == 4 and s[PREV_MULT] == 1
    s = step(mrt, INTERACTIVE, s, 0.89, 4100.0)                           # kept; nothing narrower, nothing more
    assert s[:2] == [8, 2] and s[SETTLED] == 1


def test_no_room_at_all_settles(mrt):
    s = step(mrt, C5_EIGHTH, start(mrt, C5_EIGHTH), 0.70, 30.0)     # starts at 8 x 2: nothing narrower, nothing more
    assert s[:2] == [8, 2] and s[SETTLED] == 1
    two = slots(C2, 2)
    s = step(mrt, two, start(mrt, two), 0.50, 10.0)                 # a process that runs two frames at a time
    assert s[:2] == [1, 1] and s[SETTLED] == 1


def test_a_launch_is_never_narrower_than_the_frames_that_share_the_chip(mrt):
    eight = [8, 2, 0, 1, 0, 1, 0.0]
    assert mrt.width_policy(2, C5_EIGHTH, eight, 0) == 1       # a caller that waits for every frame: the whole chip
    assert mrt.width_policy(2, C5_EIGHTH, eight, 1) == 2
    assert mrt.width_policy(2, C5_EIGHTH, eight, 7) == 8       # a full pipeline: the setting
    assert mrt.width_policy(2, C5_EIGHTH, eight, 15) == 8
    assert mrt.width_policy(2, C3, [1, 1, 0, 1, 0, 1, 0.0], 5) == 1
