"""CPU only: what the REFERENCE's demodulate() does when inputs fail (src/rtl_airband.cpp:377-391) -- the behaviour the reference-side
shim (integration/demod_hip.cpp) has to reproduce and tests/test_dropin_shim.py compares it with on the GPU.  Pins the harness's
reading of the reference: a failed device is taken out (disable_device_outputs, devices_running--), the others continue, and with
none left the thread sets do_exit and returns."""
import numpy as np
import pytest

import helpers
import pyref

INPUT_DISABLED = 5  # input_state_t (src/input-common.h:34)


@pytest.mark.skipif(not pyref.have_ref(True), reason="oracle/_ref not built (needs /root/reference)")
def test_reference_takes_failed_devices_out_and_exits_with_the_last(pkg, built):
    n_dev, n_batches, wave_rate = 3, 5, 16000
    devices, carriers = helpers.plan_devices(n_dev, True, None)
    nbytes = helpers.stream_bytes(n_batches, wave_rate) + 4 * 640
    iq = [pkg.siggen.generate_u8(d, 0, nbytes // 2, carriers) for d in range(n_dev)]
    r = pyref.run_reference_all(devices, iq, n_batches, nfm=True, fail_after=[None, 2, None], end_of_streams=True)
    assert r["batches"] == [5, 2, 5]
    assert r["outputs_disabled"] == [0, 1, 0] and r["devices_running"] == 2 and r["input_state"][1] == INPUT_DISABLED
    assert r["exited_on_its_own"] and r["devices_running_at_exit"] == 0 and r["outputs_disabled_at_exit"] == [1, 1, 1]
    whole = pyref.run_reference_all(devices, iq, n_batches, nfm=True)
    assert np.array_equal(r["axc"][0], whole["axc"][0]) and np.array_equal(r["waveout"][2], whole["waveout"][2])  # the others were not disturbed
