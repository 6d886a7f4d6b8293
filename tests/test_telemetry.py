"""CPU: the clock / power telemetry of bench.py (tools/telemetry.py -- measurement tooling, not part of the product package)
on a scripted source, the no-source case, and a sampler child that stops answering."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from telemetry import Telemetry, open_source  # noqa: E402


class _Scripted:
    name = "scripted"

    def __init__(self):
        self.n = 0

    def sample(self):
        self.n += 1
        # 1000 W for the first half of the samples, 1400 W afterwards; the energy counter advances by exactly power x time
        p = 1000.0 if self.n <= 10 else 1400.0
        return dict(sclk=2000.0 + self.n, sclk_xcd=[2000.0 + self.n] * 8, power=p, energy=1e9 + self.n * (1200.0 * 0.005 / 15.259e-6),
                    fw_ts=5e8 + self.n * (0.005 / 1e-8), temp=50.0, mclk=2000.0, activity=99.0, throttle=0)

    def cap(self):
        return 1400.0

    def close(self):
        pass


def test_summary_of_a_scripted_source():
    src = _Scripted()
    with Telemetry(source=src, hz=400.0) as t:
        time.sleep(0.08)
    s = t.summary()
    assert s["source"] == "scripted" and s["samples"] >= 12 and s["power_cap_w"] == 1400.0
    assert 2000.0 < s["sclk_mhz_min"] <= s["sclk_mhz_mean"] <= s["sclk_mhz_max"]
    assert 1000.0 <= s["power_w_mean"] <= 1400.0 and s["power_w_max"] == 1400.0
    assert len(s["sclk_mhz_mean_per_xcd"]) == 8
    # the firmware counters advance by 1200 W x 5 ms per sample: the energy-derived power is that, whatever the sampled values say
    assert abs(s["energy_power_w"] - 1200.0) < 1.0
    assert s["throttle_status_or"] == "0x0"


def test_a_box_without_any_source_reports_that_and_does_not_raise():
    src, why = open_source(0)
    if src is not None:          # a GPU box: nothing to check here
        src.close()
        return
    s = Telemetry(0).start().stop().summary()
    assert s["source"] is None and "amdsmi" in s["error"]


def test_not_available_fields_do_not_poison_the_means():
    class Holes(_Scripted):
        def sample(self):
            d = super().sample()
            if self.n % 2:
                d["power"], d["sclk"], d["sclk_xcd"] = None, None, None
            d["energy"] = None
            return d
    with Telemetry(source=Holes(), hz=400.0) as t:
        time.sleep(0.05)
    s = t.summary()
    assert s["power_w_mean"] is not None and s["sclk_mhz_mean"] is not None and "energy_power_w" not in s


def test_a_child_that_misses_an_answer_is_dropped_for_every_instance(monkeypatch):
    """The sampler child is shared per device; one that timed out must not hand its late reply to the next instance."""
    class Mute:
        closed = 0

        def ask(self, cmd, timeout=5.0):
            return None

        def cap(self):
            return None

        def close(self):
            Mute.closed += 1

    key = ("child", 0, 200.0)
    monkeypatch.setitem(Telemetry._shared, key, (Mute(), []))
    t = Telemetry(0)
    assert t.child is Telemetry._shared[key][0]
    t.start()
    assert t.child is None and key not in Telemetry._shared and Mute.closed == 1
    assert t.stop().summary()["source"] is None
