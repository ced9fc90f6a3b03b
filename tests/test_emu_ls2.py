"""CPU-only checks of the long-stream front end (csrc/rfid_ls2.hpp + the launch list of csrc/rfid_ls2_enqueue.hpp): the
UNMODIFIED kernel source and launch sequence run on the lock-step wave emulator (tests/wave_emu, test infrastructure)
and must reproduce the oracle's sequential gate scan bit for bit -- window starts, types, dc_est at every opening,
everything the decoder derives from them -- and an independent restatement of the avg_ampl recurrence at every cut.
The front end never assumes: a piece whose run does not PROVABLY cover its true start value is run again, cuts that
turn out not to be idle are withdrawn, and when rounds run out the sequential scan takes over; every one of those paths
is driven here.  (The real parity gate is on the MI355X: tests/test_gpu_*.py.)"""
import numpy as np
import pytest

import parity


def avg_reference(y):
    """gate_impl.cc:130-133 restated in numpy binary32 scalars: avg_ampl BEFORE every sample (and after the last)"""
    amp = np.sqrt(y.real.astype(np.float64) ** 2 + y.imag.astype(np.float64) ** 2).astype(np.float32)
    ring = np.zeros(100, np.float32)
    avg = np.float32(0.0)
    out = np.empty(len(y) + 1, np.float32)
    hundred = np.float32(100.0)
    for i in range(len(y)):
        out[i] = avg
        wi = i % 100
        avg = np.float32(avg + np.float32(np.float32(amp[i] - ring[wi]) / hundred))
        ring[wi] = amp[i]
    out[len(y)] = avg
    return out


def _check(emu_mod, oracle_mod, raw2d, lens=None, cfg_kw=None, check_avg=True, expect_ok=None, **kw):
    cfg_kw = cfg_kw or {}
    r = emu_mod.ls2_process(raw2d, lens=lens, **cfg_kw, **kw)
    B = raw2d.shape[0]
    for b, (wb, rb, sb) in enumerate(parity.split_by_stream(r["windows"], r["results"], r["scores"], B)):
        n = raw2d.shape[1] if lens is None else lens[b]
        o = oracle_mod.run_trace(raw2d[b, :n], oracle_mod.config(**cfg_kw))
        parity.compare_trace(wb, rb, sb, r["stats"][b], o)
        if check_avg and r["ok"]:
            ref = avg_reference(oracle_mod.fir(raw2d[b, :n]))
            for pc in r["pieces"]:
                if pc[0] == b:
                    assert pc[3:4].view(np.uint32)[0] == ref[pc[1]].view(np.uint32), ("avg_ampl at a cut", b, pc)
    if expect_ok is not None:
        assert r["ok"] == expect_ok, r["ctl"]
    return r


LS2_AVG_ROUNDS = 11      # rfid_ls2.hpp: re-run rounds of avg_ampl a pass enqueues at most


FUSED = pytest.mark.parametrize("fused", [False, True], ids=["y-given", "fused-first-pass"])
# fused: the first avg_ampl pass runs the matched filter itself from the raw samples and finds its own piece boundaries
# (ls2_front_kernel: what rfid_batch_process takes for fresh traces); else y is filtered first and the cut searches run on it
# (the streaming / look-ahead form).  Whole units in the dc_est stage when fused (its units' table holds no pieces).


@FUSED
def test_ls2_matches_oracle_and_avg_recurrence(emu_mod, oracle_mod, synth_mod, fused):
    """Two traces of different carrier level / phase / noise, cut into ~10 pieces each: accepted, and identical to the
    sequential scan; avg_ampl at every cut equals the in-order recurrence."""
    rng = np.random.default_rng(3)
    ts = []
    for k, (sigma, leak) in enumerate([(0.01, 14.9 * np.exp(0.91j)), (0.03, 3.8 * np.exp(3.4j))]):
        ts.append(synth_mod.make_trace(n_rounds=16, sigma=sigma, seed=200 + k, leak=leak, t1_jitter_raw=4).samples)
    L = min(map(len, ts))
    r = _check(emu_mod, oracle_mod, np.stack([t[:L] for t in ts]), expect_ok=1, fused=fused)
    c = r["ctl"]
    assert c["n_pieces"] >= 60 and c["n_units"] == c["n_heads"] >= 12 and c["n_windows"] == len(r["windows"])
    # dc_est: every unit from 64 neighbouring starts; the first round's centres (ring means) are off by the rounding drift, the
    # second round's are the chain's predictions: two rounds, nothing left for the finishing walk
    assert c["n_units"] == c["n_dc_pieces"] and c["dc_rounds"] <= 3 and c["dc_finished"] == 0, c
    if fused:
        assert all(pc[1] % 64 == 0 for pc in r["pieces"])        # its pieces lie on block boundaries
    del rng


@FUSED
def test_ls2_rerun_rounds_are_exercised(emu_mod, oracle_mod, synth_mod, fused):
    """Noise of 8 % of the carrier: dc_est's imaginary part (25 sin 0.7 = 16.1) hovers ACROSS 16.0 all along the trace -- what a
    start value does to a unit's end is then no shift any more, nothing about a shifted start is provable, and until round 5 such a
    pass gave up (profiles/r06/noise_sweep.txt).  Every unit is run from 64 neighbouring starts and the chain of their tables is
    exact wherever the true start lies inside the window: units are run again (dc_reruns > 0), the pass is ACCEPTED, and the
    result is the sequential scan's."""
    t = synth_mod.make_trace(n_rounds=20, sigma=0.08, seed=9).samples
    r = _check(emu_mod, oracle_mod, t[None, :], expect_ok=1, fused=fused)
    assert r["ctl"]["dc_reruns"] > 0 and r["ctl"]["dc_rounds"] >= 2 and r["ctl"]["dc_finished"] == 0, r["ctl"]
    # one round only (the ring means, off by the drift, are all the chain gets): whatever is not settled -- everything behind
    # the first unit whose true start lies outside its window -- goes through the finishing walk, one unit after the other from
    # the proven value before it.  Still accepted, still the sequential scan (the partial fallback, not the whole-pass one)
    r = _check(emu_mod, oracle_mod, t[None, :], expect_ok=1, dc_rounds=0, fused=fused)
    assert r["ctl"]["dc_finished"] >= 3 and r["ctl"]["dc_count0"] == 0, r["ctl"]      # (the walk takes what it settles off the count)
    # the chain's second level (groups of 64 blocks of 64 units: traces of more than 4 096 idle-grid slots), the finishing walk
    # behind one round of it
    r = _check(emu_mod, oracle_mod, t[None, :], expect_ok=1, dc_two_levels=True, fused=fused)
    assert r["ctl"]["dc_reruns"] > 0 and r["ctl"]["dc_finished"] == 0, r["ctl"]
    r = _check(emu_mod, oracle_mod, t[None, :], expect_ok=1, dc_two_levels=True, dc_rounds=0, fused=fused)
    assert r["ctl"]["dc_finished"] >= 3, r["ctl"]
    # few, long pieces: avg_ampl too
    r = _check(emu_mod, oracle_mod, t[None, :], expect_ok=1, target=6, fused=fused)
    # (the fused first pass guesses with the drift its look-back has found -- on the emulator, where workgroups run one after
    # the other, the slot before has always finished: hardly a piece is left to run again)
    assert r["ctl"]["n_pieces"] <= 8 and (fused or r["ctl"]["avg_reruns"] > 0), r["ctl"]


@pytest.mark.parametrize("kw", [dict(), dict(dc_rounds=0), dict(dc_two_levels=True)], ids=["rounds", "finishing-walk", "two-levels"])
def test_ls2_dc_est_chain_over_several_blocks(emu_mod, oracle_mod, synth_mod, kw):
    """Pieces of 64 samples: 235 idle-grid slots = four blocks of the dc_est chain (most slots empty, the units spread over the
    blocks), on the trace whose dc_est hovers across 16.0: the tables' chain through empty nodes, several blocks and -- forced --
    the second level must still give the sequential scan's dc_est at every gate opening."""
    t = synth_mod.make_trace(n_rounds=20, sigma=0.08, seed=9).samples
    r = _check(emu_mod, oracle_mod, t[None, :], expect_ok=1, min_piece=64, check_avg=False, **kw)
    c = r["ctl"]
    assert c["n_units"] >= 20 and c["dc_count0"] + c["dc_finished"] > 0, c


@pytest.mark.parametrize("sigma,bias", [(0.03, 300), (0.01, 5000), (0.002, 1000)])
@pytest.mark.parametrize("kw", [dict(), dict(dc_two_levels=True), dict(dc_rounds=0), dict(dc_rounds=1)], ids=["rounds", "two-levels", "walk-only", "one-round"])
def test_ls2_dc_est_far_starts_margins_and_reruns(emu_mod, oracle_mod, synth_mod, sigma, bias, kw):
    """What a LONG trace does to the dc_est stage, on a short one: the first round's centres (the ring means) are moved `bias`
    ulps off (the rounding drift of 10^8 additions), so that every unit's true start lies far outside its 64-candidate window.
    Units whose sums keep away from binade edges are settled all the same -- candidate 32's / 33's run shifted along, as far as
    the run's margin reaches; the others are run again, centred on the chain's prediction, and the blocks' windows move to
    where the last chain found their entry values.  (This is where round 6's first GPU run went wrong: a block's table had
    missed, every unit inside it -- walked one by one -- was settled, and the finishing walk took the value the chain had
    written for the first unit BEHIND the block for exact.  It now takes the end of the settled unit before it.)"""
    t = synth_mod.make_trace(n_rounds=30, sigma=sigma, seed=11, leak=1.0 * np.exp(0.6j)).samples
    r = _check(emu_mod, oracle_mod, t[None, :], expect_ok=1, min_piece=64, check_avg=False, dc_bias=bias, **kw)
    c = r["ctl"]
    assert c["n_units"] >= 30 and c["fail"] == 0, c
    if kw.get("dc_rounds", 3) >= 1 and "dc_two_levels" not in kw:
        assert c["dc_finished"] == 0 and c["dc_rounds"] <= 3, c          # (the second round settles everything away from binade edges)


@FUSED
def test_ls2_carrier_at_a_power_of_two(emu_mod, oracle_mod, synth_mod, fused):
    """A carrier whose filtered amplitude is exactly 16.0: avg_ampl hovers at a binade edge, hardly any run is provable
    (pieces start in one binade and end in the other, partial sums sit next to the edge) and the chain only advances by
    runs from exact or neighbouring starts, a few pieces per round.  Whatever the front end then does -- settle after
    many rounds (few, long pieces), or give up and leave the trace to the sequential scan -- the result must be the
    sequential scan's, and avg_ampl at every cut the in-order recurrence's (a piece accepted with a shift scaled by the
    wrong binade's ulp would show here and nowhere else)."""
    t = synth_mod.make_trace(n_rounds=12, sigma=0.01, seed=5).samples
    t = (t * np.complex64(0.64)).astype(np.complex64)
    r = _check(emu_mod, oracle_mod, t[None, :], expect_ok=1, target=8, min_piece=2048, fused=fused)
    assert (r["ctl"]["avg_rounds"] >= 4 or fused) and r["ctl"]["n_pieces"] >= 5, r["ctl"]
    # (many short pieces: the y-given form runs out of rounds here; the fused first pass's guesses -- see above -- may settle it)
    r = _check(emu_mod, oracle_mod, t[None, :], expect_ok=None if fused else 0, fused=fused)
    assert (fused or r["ctl"]["avg_count%d" % LS2_AVG_ROUNDS] > 0) and r["ctl"]["n_pieces"] > 20, r["ctl"]


@FUSED
@pytest.mark.parametrize("fsm_lanes", [False, True])
def test_ls2_ragged_batch_collisions_and_limits(emu_mod, oracle_mod, synth_mod, fsm_lanes, fused):
    """Per-trace lengths (one trace cut inside a slot, one too short to be cut, one empty), FIXED_Q = 2 collisions.
    (fsm_lanes: the state machine in its one-lane-per-unit form, which the library takes on long passes.)"""
    kw = dict(fixed_q=2, tag_ids=(0x11, 0x22, 0x33), sigma=0.01, t1_jitter_raw=5)
    traces = [synth_mod.make_trace(n_rounds=r, seed=900 + r, **kw).samples for r in (4, 3, 1)]
    L = max(map(len, traces))
    raw = np.zeros((4, L), dtype=np.complex64)
    lens = []
    for i, t in enumerate(traces):
        raw[i, : len(t)] = t
        lens.append(len(t))
    lens.append(0)
    lens[1] -= 12345
    r = _check(emu_mod, oracle_mod, raw, lens=lens, cfg_kw=dict(fixed_q=2), expect_ok=1, fsm_lanes=fsm_lanes, fused=fused)
    assert r["stats"][3]["n_windows"] == 0


@FUSED
@pytest.mark.parametrize("fsm_lanes", [False, True])
def test_ls2_cuts_that_are_not_idle_are_withdrawn(emu_mod, oracle_mod, synth_mod, fsm_lanes, fused):
    """Cut points forced to arbitrary places -- inside reader commands, inside open windows, right behind a window: the
    state machine's end state does not meet the idle state assumed at the next cut (or the dc ring is not the last 48
    samples), the pieces are appended to their predecessors (n_units < n_pieces) and scanned through; avg_ampl needs no
    idle point at all.  Still the sequential scan, bit for bit."""
    t = synth_mod.make_trace(n_rounds=8, sigma=0.02, seed=77, t1_jitter_raw=3).samples
    o = oracle_mod.run_trace(t)
    n = len(t) // 5
    Pc = 4 * 512
    cand = []
    for k, (p, ty) in enumerate(zip(o.open_idx, o.dumps["type"])):
        wlen = 1370 if ty else 250
        cand += [int(p) + 100, int(p) - 30, int(p) + wlen + 2, int(p) - 400, int(p) + wlen + 47]
    cuts, seen = [], set()
    for c in cand:                       # a forced cut stands for the grid point right before it
        J = c // Pc
        if 1 <= J and c % Pc < Pc // 2 and J not in seen and c < n - 64:
            seen.add(J)
            cuts.append(c)
    assert len(cuts) >= 6
    r = _check(emu_mod, oracle_mod, t[None, :], expect_ok=1, cuts=sorted(cuts), fsm_lanes=fsm_lanes, fused=fused)
    c = r["ctl"]
    assert c["n_heads"] == len(cuts) + 1 and c["n_units"] < c["n_heads"] and c["fsm_rounds"] >= 2, c


@pytest.mark.parametrize("fsm_lanes", [False, True])
def test_ls2_streaming_form_carries_the_gate_state(emu_mod, oracle_mod, synth_mod, fsm_lanes):
    """The form rfid_stream_work uses: a call processes up to its last idle cut (hold_last), leaves the gate state there
    (rings rebuilt from the samples, recurrences from the chains) and the next call starts from it.  Two calls over one
    trace give the windows of the sequential scan over the whole trace."""
    t = synth_mod.make_trace(n_rounds=14, sigma=0.01, seed=123).samples
    o = oracle_mod.run_trace(t)
    state = np.zeros(emu_mod.lib().emu_gate_state_size(), dtype=np.uint8)
    half = (len(t) // 2) // 5 * 5
    r1 = emu_mod.ls2_process(t[None, :half], state=state, hold_last=True, fsm_lanes=fsm_lanes)
    assert r1["ok"] == 1 and 0 < r1["consumed"] < half // 5
    c1 = r1["consumed"]
    # the second call: 25 raw samples early, so that the matched filter has its history (the 5 leading outputs are skipped)
    r2 = emu_mod.ls2_process(t[None, 5 * c1 - 25:], state=state, hold_last=False, y_skip=5, fsm_lanes=fsm_lanes)
    assert r2["ok"] == 1
    k = len(r1["windows"])
    assert k > 4 and len(r2["windows"]) > 4
    assert np.array_equal(np.concatenate([r1["windows"]["start"], r2["windows"]["start"] + c1]), o.open_idx)
    assert np.array_equal(np.concatenate([r1["windows"]["type"], r2["windows"]["type"]]), o.dumps["type"])
    # dc_est is a sum over the whole past: the second call's values prove the carried rings and recurrences
    for f, ref in (("dc_re", o.dc.real), ("dc_im", o.dc.imag)):
        got = np.concatenate([r1["windows"][f], r2["windows"][f]])
        assert np.array_equal(got.view(np.uint32), ref.astype(np.float32).view(np.uint32)), f
    # ... and so does avg_ampl at the second call's cuts
    ref = avg_reference(oracle_mod.fir(t))
    for pc in r2["pieces"]:
        assert pc[3:4].view(np.uint32)[0] == ref[c1 + pc[1]].view(np.uint32), pc


def test_chain_add_auto2_equals_two_sequential_sums(emu_mod):
    """The in-order sums from two carries a few ulps apart in ONE pass (one scan while their distance is even or the step has no
    tie, two at the tie that makes it even): each must equal the plain sequential binary32 sum from its carry -- ties, binade
    edges, zeros, both signs."""
    rng = np.random.default_rng(11)
    n_scanned = 0
    for trial in range(600):
        carry = np.float32(rng.choice([23.456789, -19.12345, 31.99999, 16.000002, 0.0, 1e-30, -15.99999, 3.0e5, 25.0, 8.5, -0.75, 15.999999]))
        scale = float(rng.choice([1e-4, 1e-3, 1e-2, 0.3, 1e-8, 1e3]))
        x = (rng.standard_normal(64) * scale).astype(np.float32)
        kind = rng.integers(0, 4)
        if kind == 1:
            x[::3] = np.ldexp(rng.integers(-7, 8, len(x[::3])).astype(np.float32) + 0.5, -19)   # half-ulp at 16..32: ties
        if kind == 2:
            x[rng.integers(0, 64)] = 0.0
            x[5] = -0.0
        # the second carry: one ulp above the first (a run's two variants at its start), or what ties leave of that: equal, two
        # apart (one scan serves both while the distance is even), and other small distances
        cb = carry
        for _ in range((1, 1, 2, 0, 3, 4)[trial % 6]):
            cb = np.nextafter(cb, np.float32(np.inf))
        if carry == 0 and cb == 0:
            cb = np.float32(np.nextafter(np.float32(0), np.float32(1)))
        oa, ob, sc = emu_mod.chain_scan2(x, float(carry), float(cb))
        n_scanned += sc
        for c0, got in ((carry, oa), (np.float32(cb), ob)):
            acc = np.float32(c0)
            for i in range(64):
                acc = np.float32(acc + x[i])
                assert got[i].view(np.uint32) == acc.view(np.uint32), (trial, float(c0), i)
    assert n_scanned > 120


@FUSED
@pytest.mark.parametrize("kw", [dict(max_num_queries=7), dict(number_unique_tags=1), dict(max_num_queries=2, number_unique_tags=2)])
def test_ls2_statistics_from_summaries_with_the_terminated_cut_off(emu_mod, oracle_mod, synth_mod, kw, fused):
    """The statistics kernel on the decoder's one-word summaries (what the library does for few, long traces; the emulated
    long-stream chain runs it so) with the reader's stop conditions inside the trace: the queries limit (the cut-off index is
    searched in the summaries), the distinct-tag limit, both."""
    t = synth_mod.make_trace(n_rounds=14, sigma=0.01, seed=321, fixed_q=1, tag_ids=(0x31, 0x52)).samples
    assert oracle_mod.run_trace(t, oracle_mod.config(fixed_q=1)).state.n_unique_tags == 2     # (the limits below do cut the run)
    r = emu_mod.ls2_process(t[None, :], fixed_q=1, fused=fused, **kw)
    assert r["ok"] == 1
    o = oracle_mod.run_trace(t, oracle_mod.config(fixed_q=1, **kw))
    st = r["stats"][0]
    assert st["status"] == o.state.status == 1
    assert 0 < st["n_windows_used"] == o.n_windows < st["n_windows"]
    for k in ("n_queries_sent", "cur_inventory_round", "cur_slot_number", "n_epc_correct", "n_unique_tags"):
        assert st[k] == getattr(o.state, k), k
    assert np.array_equal(st["tag_reads"], np.array(o.state.tag_reads[:], dtype=np.int32))


@FUSED
@pytest.mark.parametrize("seed,scale,kw", [(2, 0.64, {}), (2, 0.32, {}), (6, 0.64, dict(target=8, min_piece=2048)), (6, 0.6401, {})])
def test_ls2_exact_end_put_into_a_function_serves_its_own_start_only(emu_mod, oracle_mod, synth_mod, seed, scale, kw, fused):
    """Carriers at a power of two again, the seeds on which the chain once ACCEPTED wrong starts: a piece run from six
    neighbouring starts has its exact end for the start the chain landed on put into its function (Ls2Aff); when the next
    chain lands on another start of the same parity -- the piece's own, D = 0 -- that entry is not this start's end.  The
    pass reported success with up to 33 piece starts two ulps off.  Now the plain function comes back and the chain is
    redone: every start equals the in-order recurrence, windows and scores the oracle's."""
    t = synth_mod.make_trace(n_rounds=8, sigma=0.01, seed=seed).samples
    t = (t * np.complex64(scale)).astype(np.complex64)
    r = _check(emu_mod, oracle_mod, t[None, :], fused=fused, **kw)      # (checks avg_ampl at every cut whenever the pass was accepted)
    if not fused:      # (the seeds were picked on the unfused pieces; the fused pass cuts elsewhere and may settle or give up)
        assert r["ok"] == 1, r["ctl"]


@pytest.mark.parametrize("seed,lanes", [(7002, False), (7034, True)])
def test_ls2_streaming_form_when_the_second_call_gives_up(emu_mod, oracle_mod, synth_mod, seed, lanes):
    """The streaming form with a carrier at a power of two: the first call's front end is accepted and carries the gate state,
    the second call's runs out of rounds and the sequential scan takes the call from the carried state (its windows
    numbered from 0, as the front end numbers a call's windows): together the windows and dc_est values of the sequential
    scan over the whole trace."""
    t = synth_mod.make_trace(n_rounds=12, sigma=0.01, seed=seed, t1_jitter_raw=2).samples
    t = (t * np.complex64(0.64)).astype(np.complex64)
    o = oracle_mod.run_trace(t)
    state = np.zeros(emu_mod.lib().emu_gate_state_size(), dtype=np.uint8)
    cutp = (len(t) // 2) // 5 * 5
    r1 = emu_mod.ls2_process(t[None, :cutp], state=state, hold_last=True, fsm_lanes=lanes)
    if r1["ok"] != 1 or r1["consumed"] <= 0:
        pytest.skip("the first call was not accepted on this seed")
    c1 = r1["consumed"]
    r2 = emu_mod.ls2_process(t[None, 5 * c1 - 25:], state=state, hold_last=False, y_skip=5, fsm_lanes=lanes, generous=False)
    assert r2["ok"] == 0, r2["ctl"]
    assert np.array_equal(np.concatenate([r1["windows"]["start"], r2["windows"]["start"] + c1]), o.open_idx)
    assert np.array_equal(np.concatenate([r1["windows"]["type"], r2["windows"]["type"]]), o.dumps["type"])
    for f, ref in (("dc_re", o.dc.real), ("dc_im", o.dc.imag)):
        got = np.concatenate([r1["windows"][f], r2["windows"][f]])
        assert np.array_equal(got.view(np.uint32), ref.astype(np.float32).view(np.uint32)), f
