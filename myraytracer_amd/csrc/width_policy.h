// The launch-width controller's POLICY, free of HIP and of the clock: what api.cpp's frame loop (redraw_frames) decides
// with, and what tests/test_width_policy.py drives with synthetic measurement windows on the CPU (through
// mrt_debug_width_policy, include/myraytracer_amd_debug.h).  Scheduling only -- no decision here can change an image.
//
// The reference has one schedule: one full-screen draw per State::redraw, one frame after the other (lib.rs:241-307).  Here a
// frame is launched on 1 / div of the persistent waves the chip holds and max(2, div) x mult frames are in flight:
// narrow launches pack the lanes better (more pixels per lane in sequence) at the price of latency, and do not always pay, so
// a narrower width (or, where a frame has too few tiles to be launched any narrower, twice the frames in flight) is TRIED while
// the measured lane utilisation is low, and kept only if the measured frame rate rises by 3 %.
#pragma once
#include <stdint.h>

namespace mrt {

struct WidthWorkload {
    uint32_t n_tiles;       // 8x8 tiles of the (shard of the) frame
    uint32_t n_waves;       // persistent waves of a full-width launch
    uint32_t max_slots;     // most frames that can be in flight (mrt_ctx::kMaxFrameSlots, capped by the hardware queues)
                            // (a launch is never narrower than 1 / kMaxWidthDiv, whatever max_slots)
    uint32_t spp;           // samples per pixel and frame
    uint32_t n_members;     // member slots of the scene's hierarchy (> 1,024: the large-scene kernels)
    uint32_t counter;       // counter-RNG mode
};

constexpr uint32_t kMaxWidthDiv = 8;    // the narrowest launch: an eighth of the persistent waves

struct WidthState {
    uint32_t div = 0, mult = 1;             // div 0 = not chosen yet for the current workload
    uint32_t prev_div = 0, prev_mult = 1;   // prev_div != 0: a trial is running; what it would return to
    uint32_t low_windows = 0;               // consecutive windows between the two utilisation thresholds (a trial takes two)
    uint32_t settled = 0;                   // no further trials for this workload
    double prev_rate = 0.0;                 // frames / s measured at (prev_div, prev_mult)
};

// one measurement window: lane utilisation of the frames at the current setting, their rate on the host's clock
struct WidthWindow { double util, rate; };

inline uint32_t width_min_u32(uint32_t a, uint32_t b) { return a < b ? a : b; }
inline uint32_t width_max_u32(uint32_t a, uint32_t b) { return a > b ? a : b; }

// frames in flight of a setting
inline uint32_t width_frames_in_flight(uint32_t div, uint32_t mult, uint32_t max_slots) {
    return width_min_u32(width_max_u32(2u, div) * width_max_u32(mult, 1u), width_max_u32(max_slots, 1u));
}

// What is known up front.
//  * A pixel-starved launch of long chains (fewer than two pixels per lane the chip holds, one sequential chain of >= 64
//    samples each: an 8-GPU share of C5) lasts as long as its heaviest pixel while most of its waves are done far earlier, and
//    a wave's iteration takes the same time at 1 to 4 waves per SIMD: eight frames at a time, each on an eighth of the waves
//    (C5's 1/8 share, one mrt_redraw per frame: 885 Msamples/s at 0.41 lane utilisation with 2 frames in flight on all waves
//    -> 3,090 at 0.92 with 8) -- and, where the process has the hardware queues, eight MORE queued behind them: frames end out
//    of order, a frame slot is reused only when ITS frame has ended, and between a launch's end and the next launch on its
//    slot a sixth of the chip's wave slots stood empty (profiles/r05_shard_occupancy.txt); with twice the launches the chip
//    holds, every workgroup that ends is replaced at once by one that waits (3,090 -> 3,705).
//  * Chains of a few bounces (the reference's default: ONE sample per frame): a frame is bound by its longest path -- up to
//    ray_depth wave-iterations in sequence -- not by throughput, and every iteration is shorter with fewer resident waves: a
//    quarter of the waves, four frames side by side (the trials go on from there: 1080p at 1 spp ends at an eighth, sixteen
//    in flight: 7,150 -> 9,810 Msamples/s).
//  * Frames with tiles to spare (three per persistent wave and more: six per wave of a half-width launch): a half, and -- the
//    rule for every narrow launch -- TWICE the
//    frames the chip holds.  Frames end out of order and a slot is reused only when its own frame has ended, so with exactly
//    as many launches as fit, wave slots stand empty between a launch's end and the next launch on its slot; a queued launch
//    takes every workgroup slot the moment it frees (profiles/r05_schedule_sweep.txt: C5 4,134 -> 4,245 Msamples/s, its 1/2,
//    1/4, 1/8 shares 4,119 / 3,975 / 3,193 -> 4,223 / 4,101 / 3,814, C3 12,645 -> 12,851, C4's 1/8 share 12,157 -> 12,517, C2
//    13,051 -> 13,496; every (div, 2) measured beats its (div, 1); C4's 1/8 share, 3.2 tiles per wave: (1, 1) 12,195, (2, 2)
//    12,457, (4, 2) 12,185).
inline uint32_t width_mult_for(uint32_t div, uint32_t max_slots) { return (div >= 2u && 2u * div <= max_slots) ? 2u : 1u; }
inline void width_policy_start(WidthState& s, const WidthWorkload& w) {
    const uint32_t slots = width_max_u32(w.max_slots, 2u), narrowest = width_min_u32(kMaxWidthDiv, slots);
    const bool starved = !w.counter && w.spp >= 64u && (uint64_t)w.n_tiles < 2ull * w.n_waves && w.n_tiles > w.n_waves / narrowest;
    const bool short_chains = w.spp < 4u && (uint64_t)w.n_tiles * 4u >= 2ull * w.n_waves;
    s = WidthState();
    // (... a quarter for large scenes, whose pixels' chains differ 10 x: C5's 1/2 share 4,121 Msamples/s at a half x 2, 4,447 at a
    // quarter x 2; C5 and scenes of 1,297 to 4,901 spheres the same within 1 % either way: profiles/r05_schedule_sweep.txt)
    const uint32_t spare = (w.n_members > 1024u && slots >= 8u) ? 4u : 2u;
    s.div = starved ? narrowest : short_chains ? width_min_u32(4u, slots) : (!w.counter && w.n_tiles >= 3u * w.n_waves) ? spare : 1u;
    if (!short_chains) s.mult = width_mult_for(s.div, slots);
}

// A window has closed: ends a running trial (kept only if the rate rose by 3 %, else the previous setting returns and stays),
// then decides whether another one starts.  Below 0.90 one window is evidence enough (a pixel-starved share reads 0.4 to 0.8);
// between 0.90 and 0.95 a trial takes two consecutive windows (a window's utilisation scatters by a few per cent around the
// workload's own -- C3: 0.93 to 1.02 around 0.970); at 0.95 and above, or with no room left, the setting stays.
// Narrower only while a launch's waves still get at least two tiles each; else twice the frames in flight.
// (Round 5 also gave every well-utilised setting ONE exploratory trial of the next narrower width -- C4's 1/8 share reads 0.96
// at full width and renders 3 % more on a quarter of the waves, eight in flight.  Removed: a gain of 3 % cannot be told from
// the transient after the change of setting within a few dozen frames -- C3 was "measured" 4 % faster at a quarter width, where
// it renders 2 % less -- while the trials below decide on gains of 10 % and more.  Such settings are found offline instead:
// scripts/settle_schedules.py --sweep, profiles/schedules.json, mrt_set_schedule_hint.)
inline void width_policy_step(WidthState& s, const WidthWorkload& w, const WidthWindow& m) {
    uint32_t next_div = s.div, next_mult = s.mult;
    if (s.prev_div != 0u) {
        if (m.rate < 1.03 * s.prev_rate) { next_div = s.prev_div; next_mult = s.prev_mult; s.settled = 1u; }
        s.prev_div = 0u;
    }
    if (!s.settled) {
        const uint32_t in_flight = width_max_u32(2u, s.div) * s.mult;
        const uint32_t cand = s.div == 1u ? 4u : s.div * 2u;
        const bool can_narrow = cand <= width_min_u32(kMaxWidthDiv, w.max_slots) && (uint64_t)w.n_tiles * cand >= 2ull * w.n_waves && w.n_waves >= cand;
        const bool can_add = in_flight * 2u <= w.max_slots;
        if (m.util >= 0.90 && m.util < 0.95 && (can_narrow || can_add) && ++s.low_windows < 2u) {
            // measure again at the same setting
        } else if (m.util < 0.95 && (can_narrow || can_add)) {
            s.low_windows = 0u;
            s.prev_div = s.div;
            s.prev_mult = s.mult;
            s.prev_rate = m.rate;
            if (can_narrow) { next_div = cand; next_mult = width_mult_for(cand, w.max_slots); } else next_mult = s.mult * 2u;
        } else {
            s.settled = 1u;
        }
    }
    s.div = next_div;
    s.mult = next_mult;
}

// The share of the waves ONE launch gets.  The setting assumes the caller keeps its frames in flight; a caller that waits for
// every frame (a viewer that reads the framebuffer back, or gathers and presents, after each redraw) has the chip to itself
// and would run a single frame on 1 / div of it -- up to div pixels per lane in sequence, div times the latency.  So a launch
// is never narrower than the frames that really share the chip: those still queued or running when it is issued, plus itself.
inline uint32_t width_launch_div(uint32_t div, uint32_t frames_running) {
    return width_max_u32(1u, width_min_u32(div, frames_running + 1u));
}

}  // namespace mrt
