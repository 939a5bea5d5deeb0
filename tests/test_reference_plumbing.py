"""CPU only: the pieces of the REFERENCE around demodulate() that the drop-in tests lean on, run with the reference's own demodulate() --
its mixer_thread() fed by the output thread's mixer_put_samples() calls (src/mixer.cpp, src/output.cpp:533-535), its file input driver
(src/input-file.cpp: BASELINE configs[0], "input-file with generate_signal IQ, CPU path, 1 dongle x 8 AM channels"), the waterfall it prints
(src/rtl_airband.cpp:632-643,663-667) -- and the device partition of the reference-side shim (integration/demod_hip.cpp), which is plain
arithmetic.  tests/test_dropin_shim.py runs the same scenarios with demodulate_hip() on the GPU and compares."""
import ctypes as C
import os

import numpy as np
import pytest

import helpers
import pyoracle
import pyref

need_ref = pytest.mark.skipif(not pyref.have_ref(False), reason="oracle/_ref not built (needs /root/reference)")

MIXER_CONNS = [(0, 0, 0, 1.0, 0.0), (1, 0, 0, 1.0, 0.0), (2, 0, 0, 0.5, 0.0),      # mixer 0: mono, three inputs
               (0, 1, 1, 2.0, -0.5), (1, 2, 1, 1.0, 0.5), (2, 5, 1, 1.0, 0.0),     # mixer 1: stereo (balances), a mono input among them
               (2, 3, 2, 1.5, 0.0)]                                                 # mixer 2: one input


def configs0_devices():
    """One dongle at 120.0 MHz with eight AM channels: the two of config/basic_multichannel.conf (119.5, 120.225 MHz) and six more."""
    chans, carriers = helpers.sg.baseline_plan(mixed=False)
    chans = [dict(c) for c in chans]
    chans[0]["frequency"] = 119_500_000
    chans[5]["frequency"] = 120_225_000
    return [dict(channels=chans)], carriers


def file_input_run(devices, path, n_batches, hip_lib, tries=4):
    """The driver appends half a ring (two batches) at a time and the demodulator never waits for its consumer (src/rtl_airband.cpp:649-654): a consumer that is
    descheduled for a few milliseconds on a busy box loses a batch (output_overrun_count) and everything behind it shifts.  That is the reference's own hand-off, not what is
    under test: the run is repeated until one comes through without an overrun."""
    for _ in range(tries):
        r = pyref.run_reference_all(devices, [np.zeros(0, np.uint8)], n_batches, nfm=False, file_inputs={0: (str(path), 8.0)}, wait_exit_s=20.0, timeout_s=60.0, hip_lib=hip_lib)
        if r["output_overruns"] == [0] and r["batches"][0] == n_batches:
            break
    return r


@need_ref
def test_partition_of_a_class_over_the_gpus():
    lib = C.CDLL(pyref.ref_lib_path(False, "patched"))
    lib.demod_hip_partition.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int)]
    lib.demod_hip_partition.restype = None
    for n, g in [(262144, 8), (65536, 8), (7, 8), (8, 8), (9, 8), (1, 1), (5, 2), (1000, 3)]:
        first = (C.c_int * (g + 1))()
        lib.demod_hip_partition(n, g, first)
        f = list(first)
        assert f[0] == 0 and f[-1] == n and all(a <= b for a, b in zip(f, f[1:]))          # contiguous, complete
        sizes = [b - a for a, b in zip(f, f[1:])]
        assert max(sizes) - min(sizes) <= 1                                                # balanced
        mg = __import__("importlib").import_module("rtlsdr-airband_amd.multigpu")
        assert [mg.shard_range(n, r, g) for r in range(g)] == list(zip(f, f[1:]))          # the ranks of bench.py --gpus N take the same ranges


@need_ref
def test_reference_mixers_sum_in_input_order(pkg, built):
    n_dev, n_batches, wave_rate = 3, 6, 8000
    devices, carriers = helpers.plan_devices(n_dev, False, None)
    nbytes = helpers.stream_bytes(n_batches, wave_rate) + 4 * 640
    iq = [pkg.siggen.generate_u8(d, 0, nbytes // 2, carriers) for d in range(n_dev)]
    r = pyref.run_reference_all(devices, iq, n_batches, nfm=False, mixers=(3, MIXER_CONNS))
    assert r["batches"] == [n_batches] * n_dev and r["mix_batches"] == [n_batches] * 3
    assert [m["inputs"] for m in r["mixers"]] == [3, 3, 1] and all(m["output_overruns"] == 0 for m in r["mixers"])
    opened = 0
    for b in range(n_batches):
        left, right, sig = helpers.mixer_reference_sum(MIXER_CONNS, 3, r["waveout"][:, b], r["axc"][:, b])
        assert np.array_equal(r["mix_left"][:, b].view(np.uint32), left.view(np.uint32)), b
        assert np.array_equal(r["mix_right"][1, b].view(np.uint32), right[1].view(np.uint32)), b   # the stereo mixer's right channel
        assert np.array_equal(r["mix_axc"][:, b] != ord(" "), sig != 0), b
        opened += int(sig.sum())
    assert opened > 0


@need_ref
def test_configs0_through_the_reference_file_input(pkg, built, tmp_path):
    """BASELINE configs[0]: the reference's file input replays a file of generated I/Q (paced by speedup_factor), demodulate() on the CPU, eight AM
    channels; the oracle on the same bytes; at end of file the driver reports INPUT_FAILED and the demodulator exits (src/input-file.cpp:101-111)."""
    devices, carriers = configs0_devices()
    n_batches, wave_rate = 6, 8000
    # two batches more than are compared: the driver fails its input right after its last append (src/input-file.cpp:101-104), what the demodulator
    # has not consumed by then is lost -- the reference's own end-of-file race, kept away from the batches under test
    nbytes = helpers.stream_bytes(n_batches + 2, wave_rate)
    iq = pkg.siggen.generate_u8(0, 0, nbytes // 2, carriers)
    path = tmp_path / "dongle0.u8"
    iq.tofile(path)
    r = file_input_run(devices, path, n_batches, None)
    nb = r["batches"][0]
    assert nb == n_batches and r["output_overruns"] == [0], (nb, r["output_overruns"])
    orc = pyoracle.Oracle(devices, wave_rate=wave_rate)
    want = orc.run_device(0, iq, n_batches)
    assert np.array_equal(r["axc"][0, :nb], want["axc"][:nb])
    assert np.array_equal(r["waveout"][0, :nb].view(np.uint32), want["waveout"][:nb].view(np.uint32))
    assert (want["axc"][:nb] == ord("*")).any()
    assert r["stats"][0][0]["bin"] == 411 and r["stats"][0][5]["bin"] == 44          # 119.5 and 120.225 MHz at centre 120.0 (src/config.cpp:666-667)
    assert r["exited_on_its_own"] and r["devices_running_at_exit"] == 0 and r["input_state_at_exit"] == [5]   # INPUT_DISABLED


@need_ref
def test_reference_waterfall(pkg, built, tmp_path):
    n_dev, n_batches, wave_rate = 2, 14, 8000
    devices, carriers = helpers.plan_devices(n_dev, False, None)
    nbytes = helpers.stream_bytes(n_batches, wave_rate) + 4 * 640
    iq = [pkg.siggen.generate_u8(d, 0, nbytes // 2, carriers) for d in range(n_dev)]
    out = tmp_path / "tui.txt"
    r = pyref.run_reference_all(devices, iq, n_batches, nfm=False, tui_path=str(out))
    cells = helpers.parse_waterfall(out.read_text(errors="replace"))
    assert len(cells) == n_dev * n_batches * 8
    per_dev = {d: [c for c in cells if (c[0] - 3) // 17 == d] for d in range(n_dev)}
    for d in range(n_dev):
        rows = per_dev[d]
        assert len(rows) == n_batches * 8
        for b in range(n_batches):
            for j in range(8):
                y, x, sig, noise, sym = rows[b * 8 + j]
                assert y == d * 17 + (b % 12) + 3 and x == j * 10               # dev->row scrolls through 12 lines (src/rtl_airband.cpp:663-667)
                assert sym == chr(r["axc"][d, b, j]) and -200 < noise < 20 and -200 < sig < 20
