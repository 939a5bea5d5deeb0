"""The drop-in claim, end to end: the REAL reference harness (its own device_t / channel_t / input_t objects, circbuffer_append
on the producer side, the output thread's waveavail protocol on the consumer side) run twice -- once with the reference's
demodulate(), once built from the PATCHED reference: integration/airband_hip.patch applied to a scratch copy of the reference
sources, integration/demod_hip.cpp (the translation unit the patch adds) compiled verbatim, demodulate_hip() started instead of
demodulate().  Statistics of the second run are read through the reference's own Squelch getters, i.e. the way the stats file and
the TUI read them (src/output.cpp:617-761), fed by the patch's Squelch::mirror().  Needs the prebuilt oracle/_ref (it travels to
the GPU box) and a GPU."""
import os
import numpy as np
import pytest

import helpers
import pyref

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(not (pyref.have_ref(True) and os.path.exists(pyref.ref_lib_path(True, "patched"))), reason="oracle/_ref not built")
@pytest.mark.parametrize("mixed,wave_rate", [(False, 8000), (True, 16000)])
def test_reference_harness_with_hip_backend(pkg, built, mixed, wave_rate):
    def tweak(d, ch):
        if mixed:
            ch[3]["has_iq_outputs"] = 1
            ch[0]["bandwidth_hz"] = 8000
    n_dev, n_batches = 3, 12
    devices, carriers = helpers.plan_devices(n_dev, mixed, tweak)
    nbytes = helpers.stream_bytes(n_batches, wave_rate) + 4 * 640  # a little slack: demodulate() wants one extra hop queued
    iq = [pkg.siggen.generate_u8(d, 0, nbytes // 2, carriers) for d in range(n_dev)]
    nfm = wave_rate == 16000
    if not nfm:
        pytest.importorskip("numpy")
    ref = pyref.run_reference_all(devices, iq, n_batches, nfm=nfm)
    hip = pyref.run_reference_all(devices, iq, n_batches, nfm=nfm, hip_lib=pkg.LIB_PATH)
    assert ref["n_batches"] == n_batches and hip["n_batches"] == n_batches
    assert np.array_equal(ref["axc"], hip["axc"]), "squelch decisions differ between demodulate() and the HIP backend"
    assert (ref["axc"] == ord("*")).any()
    assert helpers.rms(ref["waveout"] - hip["waveout"]) <= 1e-4
    assert helpers.rms(ref["iq_out"] - hip["iq_out"]) <= 1e-4 * max(1.0, helpers.rms(ref["iq_out"]))
    for d in range(n_dev):
        for j in range(8):
            a, b = ref["stats"][d][j], hip["stats"][d][j]
            for k in ("open_count", "flappy_count", "ctcss_count", "no_ctcss_count", "active_counter", "bin"):
                assert a[k] == b[k], (d, j, k, a[k], b[k])
            for k in ("noise_level", "signal_level", "squelch_level", "agcavgfast"):  # through Squelch::mirror() and the getters
                assert abs(a[k] - b[k]) <= 1e-4 * max(abs(a[k]), 1e-2), (d, j, k, a[k], b[k])


INPUT_DISABLED = 5  # input_state_t (src/input-common.h:34)


def _failure_run(pkg, hip_lib, env=None):
    n_dev, n_batches, wave_rate = 3, 8, 16000
    devices, carriers = helpers.plan_devices(n_dev, True, None)
    nbytes = helpers.stream_bytes(n_batches, wave_rate) + 4 * 640
    iq = [pkg.siggen.generate_u8(d, 0, nbytes // 2, carriers) for d in range(n_dev)]
    return pyref.run_reference_all(devices, iq, n_batches, nfm=True, hip_lib=hip_lib, fail_after=[None, 3, None], end_of_streams=True, env=env)


@pytest.mark.skipif(not (pyref.have_ref(True) and os.path.exists(pyref.ref_lib_path(True, "patched"))), reason="oracle/_ref not built")
def test_one_input_at_end_of_file_then_all(pkg, built):
    """What demodulate() does when inputs stop (src/rtl_airband.cpp:377-391; the file input sets INPUT_FAILED at end of file,
    src/input-file.cpp:101-111): device 1 of 3 fails after 3 batches -> its outputs are disabled, devices_running drops to 2, the
    other two devices keep producing exactly what the reference produces; when the last input ends the demodulator sets do_exit and
    returns.  Run once with demodulate(), once with demodulate_hip() from the patched reference."""
    ref = _failure_run(pkg, None)
    hip = _failure_run(pkg, pkg.LIB_PATH)
    for r in (ref, hip):
        assert r["batches"] == [8, 3, 8]
        assert r["outputs_disabled"] == [0, 1, 0] and r["devices_running"] == 2 and r["input_state"][1] == INPUT_DISABLED
        assert r["exited_on_its_own"], "the demodulator kept spinning after every input had failed"
        assert r["devices_running_at_exit"] == 0 and r["outputs_disabled_at_exit"] == [1, 1, 1]
    for d, nb in enumerate(ref["batches"]):
        assert np.array_equal(ref["axc"][d, :nb], hip["axc"][d, :nb]), d
        assert helpers.rms(ref["waveout"][d, :nb] - hip["waveout"][d, :nb]) <= 1e-4
        for j in range(8):
            a, b = ref["stats"][d][j], hip["stats"][d][j]
            for k in ("open_count", "flappy_count", "ctcss_count", "no_ctcss_count", "active_counter"):
                assert a[k] == b[k], (d, j, k, a[k], b[k])
    assert (ref["axc"][0] == ord("*")).any() and (ref["axc"][2] == ord("*")).any()


@pytest.mark.skipif(not (pyref.have_ref(True) and os.path.exists(pyref.ref_lib_path(True, "patched"))), reason="oracle/_ref not built")
def test_shard_with_two_device_classes(pkg, built):
    """The reference takes sample rate and sample format per device (src/rtl_airband.cpp:394,402-455): one demodulate() shard with a
    2.56 MS/s u8 dongle and a 2.4 MS/s CS16 (SoapySDR) device.  The shim runs one library handle per class."""
    capi = pkg.capi
    n_batches, wave_rate = 6, 16000
    d0, iq0 = helpers.format_case(pkg, capi.SFMT_U8, 9, 2_560_000, wave_rate, 1, n_batches)
    d1, iq1 = helpers.format_case(pkg, capi.SFMT_S16, 9, 2_400_000, wave_rate, 1, n_batches, first_dongle=1)
    devices, iq = d0 + d1, [iq0[0], iq1[0].view(np.uint8)]
    ref = pyref.run_reference_all(devices, iq, n_batches, nfm=True)
    hip = pyref.run_reference_all(devices, iq, n_batches, nfm=True, hip_lib=pkg.LIB_PATH)
    assert ref["batches"] == hip["batches"] == [n_batches, n_batches]
    assert np.array_equal(ref["axc"], hip["axc"])
    for d in range(2):
        assert (ref["axc"][d] == ord("*")).any()
        assert helpers.rms(ref["waveout"][d] - hip["waveout"][d]) <= 1e-4
        for j in range(8):
            a, b = ref["stats"][d][j], hip["stats"][d][j]
            for k in ("open_count", "flappy_count", "ctcss_count", "no_ctcss_count", "active_counter", "bin"):
                assert a[k] == b[k], (d, j, k, a[k], b[k])


need_patched_am = pytest.mark.skipif(not (pyref.have_ref(False) and os.path.exists(pyref.ref_lib_path(False, "patched"))), reason="oracle/_ref not built")
need_patched_nfm = pytest.mark.skipif(not (pyref.have_ref(True) and os.path.exists(pyref.ref_lib_path(True, "patched"))), reason="oracle/_ref not built")


def _mixer_scenario(pkg, n_dev, n_batches, nfm):
    wave_rate = 16000 if nfm else 8000
    devices, carriers = helpers.plan_devices(n_dev, nfm, None)
    nbytes = helpers.stream_bytes(n_batches, wave_rate) + 4 * 640
    iq = [pkg.siggen.generate_u8(d, 0, nbytes // 2, carriers) for d in range(n_dev)]
    return devices, iq


@need_patched_am
def test_mixers_served_on_the_gpu(pkg, built):
    """The reference's mixers (mixer_t, O_MIXER outputs, mixer_thread(), src/mixer.cpp) with demodulate(), then with the HIP backend: the shim wires the
    mixers into the library (airband_hip_set_mixers from the reference's own mixinput_t values), marks them gpu_served -- mixer_put_samples() and
    mixer_thread() then leave them alone -- and publishes the sums where mixer_thread() would have.  Stereo, ampfactors, a mixer with one input."""
    from test_reference_plumbing import MIXER_CONNS
    n_dev, n_batches = 3, 6
    devices, iq = _mixer_scenario(pkg, n_dev, n_batches, False)
    ref = pyref.run_reference_all(devices, iq, n_batches, nfm=False, mixers=(3, MIXER_CONNS))
    hip = pyref.run_reference_all(devices, iq, n_batches, nfm=False, mixers=(3, MIXER_CONNS), hip_lib=pkg.LIB_PATH)
    assert hip["batches"] == [n_batches] * n_dev and hip["mix_batches"] == [n_batches] * 3
    assert [m["gpu_served"] for m in hip["mixers"]] == [1, 1, 1] and [m["gpu_served"] for m in ref["mixers"]] == [0, 0, 0]
    assert np.array_equal(ref["axc"], hip["axc"]) and np.array_equal(ref["mix_axc"], hip["mix_axc"])
    assert (hip["mix_axc"] != ord(" ")).any()
    for m in range(3):
        assert helpers.rms(ref["mix_left"][m] - hip["mix_left"][m]) <= 1e-4 * max(1.0, helpers.rms(ref["mix_left"][m]))
    assert helpers.rms(ref["mix_right"][1] - hip["mix_right"][1]) <= 1e-4 and np.abs(hip["mix_right"][1]).max() > 0
    # ... and bit for bit the reference's own summation order over the HIP backend's channel audio
    for b in range(n_batches):
        left, right, sig = helpers.mixer_reference_sum(MIXER_CONNS, 3, hip["waveout"][:, b], hip["axc"][:, b])
        assert np.array_equal(hip["mix_left"][:, b].view(np.uint32), left.view(np.uint32)), b
        assert np.array_equal(hip["mix_right"][1, b].view(np.uint32), right[1].view(np.uint32)), b


@need_patched_nfm
def test_a_shard_spread_over_two_parts(pkg, built):
    """AIRBAND_HIP_GPUS = "0,0": the shim cuts the shard's five devices into two contiguous parts (2 + 3), one library handle each -- on a node with
    several GPUs one per GPU, here both on GPU 0 -- and the mixer sums of the parts meet on the GPU (airband_hip_add_mixers; airband_hip_allreduce_mixers
    over RCCL between different GPUs).  Everything a device produces must be bit-identical to the one-handle run; mixers whose inputs lie in both parts
    too where the association order is the same, within float tolerance otherwise."""
    n_dev, n_batches = 5, 6
    devices, iq = _mixer_scenario(pkg, n_dev, n_batches, True)
    conns = [(d, 0, 0, 1.0, 0.0) for d in range(5)] + [(0, 2, 1, 2.0, -0.25), (4, 2, 1, 1.0, 0.5)] + [(3, 4, 2, 1.0, 0.0), (4, 6, 2, 0.5, 0.0)]
    one = pyref.run_reference_all(devices, iq, n_batches, nfm=True, mixers=(3, conns), hip_lib=pkg.LIB_PATH, env={"AIRBAND_HIP_GPUS": "0"})
    two = pyref.run_reference_all(devices, iq, n_batches, nfm=True, mixers=(3, conns), hip_lib=pkg.LIB_PATH, env={"AIRBAND_HIP_GPUS": "0,0"})
    for r in (one, two):
        assert r["batches"] == [n_batches] * n_dev and r["mix_batches"] == [n_batches] * 3 and [m["gpu_served"] for m in r["mixers"]] == [1, 1, 1]
    assert np.array_equal(one["axc"], two["axc"]) and (one["axc"] == ord("*")).any()
    assert np.array_equal(one["waveout"].view(np.uint32), two["waveout"].view(np.uint32))
    assert np.array_equal(one["iq_out"].view(np.uint32), two["iq_out"].view(np.uint32))
    assert one["stats"] == two["stats"]
    assert np.array_equal(one["mix_axc"], two["mix_axc"]) and (one["mix_axc"] != ord(" ")).any()
    # mixer 1: one input per part, mixer 2: both inputs in the second part -> the same additions in the same order
    for m in (1, 2):
        assert np.array_equal(one["mix_left"][m].view(np.uint32), two["mix_left"][m].view(np.uint32)), m
    assert np.array_equal(one["mix_right"][1].view(np.uint32), two["mix_right"][1].view(np.uint32))
    # mixer 0: (d0 + d1) + ((d2 + d3) + d4) against (((d0 + d1) + d2) + d3) + d4
    assert helpers.rms(one["mix_left"][0] - two["mix_left"][0]) <= 1e-6 * max(1.0, helpers.rms(one["mix_left"][0]))


@need_patched_am
def test_waterfall_with_the_hip_backend(pkg, built, tmp_path):
    """`tui` = 1: the shim prints the reference's waterfall cells (src/rtl_airband.cpp:632-643) from the mirrored statistics, '~' for
    Squelch::signal_outside_filter() included, and scrolls dev->row like demodulate() (:663-667)."""
    n_dev, n_batches, wave_rate = 2, 14, 8000
    devices, carriers = helpers.plan_devices(n_dev, False, None)
    nbytes = helpers.stream_bytes(n_batches, wave_rate) + 4 * 640
    iq = [pkg.siggen.generate_u8(d, 0, nbytes // 2, carriers) for d in range(n_dev)]
    ref = pyref.run_reference_all(devices, iq, n_batches, nfm=False, tui_path=str(tmp_path / "ref.txt"))
    hip = pyref.run_reference_all(devices, iq, n_batches, nfm=False, tui_path=str(tmp_path / "hip.txt"), hip_lib=pkg.LIB_PATH)
    a = helpers.parse_waterfall((tmp_path / "ref.txt").read_text(errors="replace"))
    b = helpers.parse_waterfall((tmp_path / "hip.txt").read_text(errors="replace"))
    assert len(a) == len(b) == n_dev * n_batches * 8
    key = lambda c: (c[0], c[1])  # the two runs interleave their devices differently (one thread walks the devices; the shim publishes a class at a time)
    for d in range(n_dev):
        ra = [c for c in a if (c[0] - 3) // 17 == d]
        rb = [c for c in b if (c[0] - 3) // 17 == d]
        assert [key(c) for c in ra] == [key(c) for c in rb]
        for x, y in zip(ra, rb):
            assert x[4] == y[4] and abs(x[2] - y[2]) <= 1 and abs(x[3] - y[3]) <= 1, (x, y)
    assert any(c[4] == "*" for c in b)
    assert np.array_equal(ref["axc"], hip["axc"])


@need_patched_am
def test_configs0_file_input_with_the_hip_backend(pkg, built, tmp_path):
    """BASELINE configs[0] with the backend switched: the reference's file input driver (src/input-file.cpp) replays generated I/Q into the device's ring,
    demodulate_hip() instead of demodulate(); end of file -> INPUT_FAILED -> the shim disables the device and, with no receiver left, exits."""
    from test_reference_plumbing import configs0_devices
    devices, carriers = configs0_devices()
    n_batches, wave_rate = 6, 8000
    iq = pkg.siggen.generate_u8(0, 0, helpers.stream_bytes(n_batches + 2, wave_rate) // 2, carriers)
    path = tmp_path / "dongle0.u8"
    iq.tofile(path)
    from test_reference_plumbing import file_input_run
    ref = file_input_run(devices, path, n_batches, None)
    hip = file_input_run(devices, path, n_batches, pkg.LIB_PATH)
    assert ref["batches"] == hip["batches"] == [n_batches] and hip["output_overruns"] == [0]
    assert np.array_equal(ref["axc"], hip["axc"]) and (hip["axc"] == ord("*")).any()
    assert helpers.rms(ref["waveout"] - hip["waveout"]) <= 1e-4
    assert hip["stats"][0][0]["bin"] == 411 and hip["stats"][0][5]["bin"] == 44
    for r in (ref, hip):
        assert r["exited_on_its_own"] and r["devices_running_at_exit"] == 0 and r["input_state_at_exit"] == [5]


@need_patched_nfm
def test_failed_inputs_with_the_shard_on_two_parts(pkg, built):
    """The end-of-file scenario of test_one_input_at_end_of_file_then_all with the shard cut into two parts (AIRBAND_HIP_GPUS = "0,0": devices {0} and {1, 2}): device 1 fails
    after three batches inside the second part, the other device of that part and the other part carry on, the last failures end the process -- and every batch is the
    one-part run's, bit for bit."""
    one = _failure_run(pkg, pkg.LIB_PATH, env={"AIRBAND_HIP_GPUS": "0"})
    two = _failure_run(pkg, pkg.LIB_PATH, env={"AIRBAND_HIP_GPUS": "0,0"})
    for r in (one, two):
        assert r["batches"] == [8, 3, 8] and r["outputs_disabled"] == [0, 1, 0] and r["devices_running"] == 2
        assert r["exited_on_its_own"] and r["devices_running_at_exit"] == 0 and r["outputs_disabled_at_exit"] == [1, 1, 1]
    for d, nb in enumerate(one["batches"]):
        assert np.array_equal(one["axc"][d, :nb], two["axc"][d, :nb])
        assert np.array_equal(one["waveout"][d, :nb].view(np.uint32), two["waveout"][d, :nb].view(np.uint32))
    assert (one["axc"][0] == ord("*")).any()


# ---- served mixers when a WHOLE part loses its devices (src/mixer.cpp:96-112 mixer_disable_input, :189-214; called for a failed device's outputs,
# src/rtl_airband.cpp:383-391): the reference keeps mixing the inputs that are left ------------------------------------------------------------------
FAKE_RCCL = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "fake_rccl", "libfake_rccl.so")
PART_CONNS = [(d, 0, 0, 1.0, 0.0) for d in range(4)] + [(0, 2, 1, 2.0, -0.25), (3, 2, 1, 1.0, 0.5)] + [(2, 4, 2, 1.0, 0.0), (3, 6, 2, 0.5, 0.0)] + [(0, 4, 3, 1.0, 0.0), (1, 6, 3, 1.5, 0.0)]
N_PART_MIXERS = 4


def _fabric_env(tmp_path, tag):
    """"0,0" as two FABRIC ranks: the shim's RCCL leg (comm_init_all, group begin / allreduce_mixers per part / group end) through the in-process stand-in
    for librccl (tests/fake_rccl/), which lets two ranks share the box's one GPU and logs every collective it completes."""
    log = tmp_path / ("fake_rccl_%s.log" % tag)
    return {"AIRBAND_HIP_GPUS": "0,0", "AIRBAND_HIP_FABRIC_PER_PART": "1", "AIRBAND_HIP_RCCL_LIB": FAKE_RCCL, "FAKE_RCCL_LOG": str(log)}, log


def _collectives(log):
    return [l for l in log.read_text().splitlines() if l.startswith("allreduce ")] if log.exists() else []


@need_patched_nfm
@pytest.mark.parametrize("dead_part", [0, 1], ids=["part0_dies", "part1_dies"])
@pytest.mark.parametrize("exchange", ["add_mixers", "fabric"])
def test_served_mixers_survive_the_loss_of_a_whole_part(pkg, built, tmp_path, dead_part, exchange):
    """Four devices in two parts ({0, 1} and {2, 3}), four served mixers: one fed by all four devices, a stereo one fed from both parts, one fed by the second part
    only, one by the first only.  After three batches BOTH devices of one part fail.  From then on that part runs no batch; the other part's devices and the mixers
    must carry on: every mixer keeps reaching CH_READY, its sum is the reference-order sum over the inputs that are left (no stale partial sums of the dead part, no
    feedback of an earlier all-reduce), the mixer fed by the dead part alone falls silent without a signal flag.  Once with the parts' sums meeting through
    airband_hip_add_mixers (two parts on one GPU), once through the RCCL leg with two ranks (the stand-in library)."""
    if exchange == "fabric" and not os.path.exists(FAKE_RCCL):
        pytest.skip("tests/fake_rccl/libfake_rccl.so not built")
    n_dev, n_batches, k_fail = 4, 8, 3
    devices, iq = _mixer_scenario(pkg, n_dev, n_batches, True)
    fail_after = [k_fail, k_fail, None, None] if dead_part == 0 else [None, None, k_fail, k_fail]
    env, log = ({"AIRBAND_HIP_GPUS": "0,0"}, None) if exchange == "add_mixers" else _fabric_env(tmp_path, "dead%d" % dead_part)
    one = pyref.run_reference_all(devices, iq, n_batches, nfm=True, mixers=(N_PART_MIXERS, PART_CONNS), hip_lib=pkg.LIB_PATH, fail_after=fail_after, env={"AIRBAND_HIP_GPUS": "0"})
    two = pyref.run_reference_all(devices, iq, n_batches, nfm=True, mixers=(N_PART_MIXERS, PART_CONNS), hip_lib=pkg.LIB_PATH, fail_after=fail_after, env=env)
    want_batches = [k_fail if f is not None else n_batches for f in fail_after]
    for r in (one, two):
        assert r["batches"] == want_batches, r["batches"]
        assert r["mix_batches"] == [n_batches] * N_PART_MIXERS, "a served mixer stopped being published: %s" % r["mix_batches"]
        assert [m["gpu_served"] for m in r["mixers"]] == [1] * N_PART_MIXERS
    live = [d for d in range(n_dev) if fail_after[d] is None]
    for d, nb in enumerate(want_batches):
        assert np.array_equal(one["axc"][d, :nb], two["axc"][d, :nb])
        assert np.array_equal(one["waveout"][d, :nb].view(np.uint32), two["waveout"][d, :nb].view(np.uint32))
    assert any((one["axc"][d, k_fail:] == ord("*")).any() for d in live), "the scenario must have signal after the failure"
    dead_only = 3 if dead_part == 0 else 2  # the mixer all of whose inputs sit in the dead part
    for b in range(n_batches):
        axc = two["axc"][:, b].copy()
        for d in range(n_dev):
            if fail_after[d] is not None and b >= k_fail:
                axc[d, :] = ord(" ")  # mixer_disable_input(): a failed device's inputs are masked out
        left, right, sig = helpers.mixer_reference_sum(PART_CONNS, N_PART_MIXERS, two["waveout"][:, b], axc)
        assert np.array_equal(two["mix_axc"][:, b] != ord(" "), sig != 0), (b, two["mix_axc"][:, b], sig)
        assert np.array_equal(one["mix_axc"][:, b], two["mix_axc"][:, b]), b
        scale = max(1.0, helpers.rms(left))
        for r in (one, two):
            assert helpers.rms(r["mix_left"][:, b] - left) <= 1e-6 * scale, "batch %d: mixer sums are not the sums over the live inputs" % b
            assert helpers.rms(r["mix_right"][1, b] - right[1]) <= 1e-6 * scale, b
        # one handle adds in the reference's order: bit for bit (a masked input is skipped, not added as zero)
        assert np.array_equal(one["mix_left"][:, b], left) and np.array_equal(one["mix_right"][1, b], right[1]), b
        if b >= k_fail:
            assert not two["mix_left"][dead_only, b].any() and two["mix_axc"][dead_only, b] == ord(" ")
            # the live part's sums arrive unchanged: zeros + x (the cleared part first or second)
            assert np.array_equal(two["mix_left"][:, b], left) and np.array_equal(two["mix_right"][1, b], right[1]), b
    assert (two["mix_axc"][:, k_fail:] != ord(" ")).any()
    if exchange == "fabric":
        lines = _collectives(log)
        assert lines and all("nranks=2" in l for l in lines), lines[:3]
        # three collectives (left, right, flags) per rank and batch
        assert len(lines) == 2 * 3 * n_batches, len(lines)


@need_patched_nfm
def test_a_shard_spread_over_two_fabric_ranks(pkg, built, tmp_path):
    """test_a_shard_spread_over_two_parts with the parts' sums meeting over the RCCL leg instead of airband_hip_add_mixers: comm_init_all over two handles,
    per batch one group with both handles' allreduce_mixers -- two ranks, through the stand-in library.  Rank order = part order, so the sums are bit for bit
    those of the add_mixers run (part 0's partial + part 1's)."""
    if not os.path.exists(FAKE_RCCL):
        pytest.skip("tests/fake_rccl/libfake_rccl.so not built")
    n_dev, n_batches = 5, 6
    devices, iq = _mixer_scenario(pkg, n_dev, n_batches, True)
    conns = [(d, 0, 0, 1.0, 0.0) for d in range(5)] + [(0, 2, 1, 2.0, -0.25), (4, 2, 1, 1.0, 0.5)] + [(3, 4, 2, 1.0, 0.0), (4, 6, 2, 0.5, 0.0)]
    env, log = _fabric_env(tmp_path, "spread")
    add = pyref.run_reference_all(devices, iq, n_batches, nfm=True, mixers=(3, conns), hip_lib=pkg.LIB_PATH, env={"AIRBAND_HIP_GPUS": "0,0"})
    fab = pyref.run_reference_all(devices, iq, n_batches, nfm=True, mixers=(3, conns), hip_lib=pkg.LIB_PATH, env=env)
    for r in (add, fab):
        assert r["batches"] == [n_batches] * n_dev and r["mix_batches"] == [n_batches] * 3 and [m["gpu_served"] for m in r["mixers"]] == [1, 1, 1]
    assert np.array_equal(add["axc"], fab["axc"]) and np.array_equal(add["waveout"].view(np.uint32), fab["waveout"].view(np.uint32))
    assert np.array_equal(add["mix_axc"], fab["mix_axc"]) and (fab["mix_axc"] != ord(" ")).any()
    assert np.array_equal(add["mix_left"], fab["mix_left"]) and np.array_equal(add["mix_right"], fab["mix_right"])
    assert np.abs(fab["mix_left"]).max() > 0 and np.abs(fab["mix_right"][1]).max() > 0
    lines = _collectives(log)
    assert len(lines) == 2 * 3 * n_batches and all("nranks=2" in l for l in lines), (len(lines), lines[:2])
