"""Clock / power telemetry of one GPU over a measured region (bench.py, tools/power_probe.py).

Box-to-box spread on MI355X is +-3..6 % (clock / power state), more than most kernel changes: a throughput
number without the shader clock and the socket power it was measured at cannot be compared across boxes.
``Telemetry`` samples them on a side thread while the region runs:

    with Telemetry(device_index=0) as t:
        ... launches ...; torch.cuda.synchronize()
    t.summary()  ->  {"sclk_mhz_mean", "sclk_mhz_min", "sclk_mhz_max", "power_w_mean", "power_w_max", "power_cap_w",
                      "energy_power_w", "samples", "hz", "source", ...}

Sources, first one that answers: (1) the ``amdsmi`` Python binding (one ``gpu_metrics`` read per sample: per-XCD
shader clocks, socket power, the firmware's energy accumulator -- the latter gives the mean power of the region
independently of the sampling rate), (2) the amdgpu hwmon files under /sys/class/drm/card*/device/hwmon, (3) one
``rocm-smi --json`` call per sample (slow: a few Hz).  Nothing here is on the product path and nothing raises: a box
without any source yields ``{"source": None, "error": ...}`` and the bench line says so.

The sampler runs in a CHILD PROCESS by default (``python telemetry.py --serve``, one per device, commands over a pipe): the
management library is then never loaded into the process that drives the GPU -- a fault inside it (one bench run of this round
died with SIGSEGV and an empty output while sampling in-process) costs the telemetry, not the measurement.  Time stamps are
CLOCK_MONOTONIC in both processes.  ``TFIMM_TELEMETRY=inproc`` samples on a thread of the calling process, ``=off`` disables it.
(The reference's own harness has neither synchronisation nor telemetry: tfimm/utils/profile.py:30-42.)
"""
import glob
import json
import os
import subprocess
import threading
import time

_NA = (None, "N/A", 0xFFFF, 0xFFFFFFFF)


def _num(v):
    """amdsmi marks fields the firmware does not fill with the type's max value or the string "N/A"."""
    if isinstance(v, bool) or v in _NA:
        return None
    if isinstance(v, (int, float)):
        return float(v)
    return None


class _AmdSmiSource:
    name = "amdsmi gpu_metrics"

    def __init__(self, device_index, bus_id=None):
        import amdsmi
        self.smi = amdsmi
        amdsmi.amdsmi_init()
        handles = amdsmi.amdsmi_get_processor_handles()
        if not handles:
            raise RuntimeError("amdsmi: no processors")
        self.h = None
        if bus_id:
            for h in handles:
                try:
                    if amdsmi.amdsmi_get_gpu_device_bdf(h).lower().endswith(bus_id.lower()):
                        self.h = h
                        break
                except Exception:  # noqa: BLE001
                    pass
        if self.h is None:
            self.h = handles[min(device_index, len(handles) - 1)]
        self.sample()          # raises here if the metrics table cannot be read

    def sample(self):
        m = self.smi.amdsmi_get_gpu_metrics_info(self.h)
        clks = [c for c in (_num(v) for v in (m.get("current_gfxclks") or [])) if c]
        sclk = (sum(clks) / len(clks)) if clks else (_num(m.get("current_gfxclk")) or _num(m.get("average_gfxclk_frequency")))
        power = _num(m.get("current_socket_power"))
        if power is None:
            power = _num(m.get("average_socket_power"))
        return dict(sclk=sclk, sclk_xcd=clks or None, power=power, energy=_num(m.get("energy_accumulator")),
                    fw_ts=_num(m.get("firmware_timestamp")), temp=_num(m.get("temperature_hotspot")),
                    mclk=_num(m.get("current_uclk")), activity=_num(m.get("average_gfx_activity")),
                    throttle=m.get("indep_throttle_status") if isinstance(m.get("indep_throttle_status"), int) else None)

    def cap(self):
        try:
            c = self.smi.amdsmi_get_power_cap_info(self.h)
            v = _num(c.get("power_cap"))
            if v is None:
                return None
            return v / 1e6 if v > 1e5 else v           # microwatts in the binding, watts in older ones
        except Exception:  # noqa: BLE001
            return None

    def close(self):
        try:
            self.smi.amdsmi_shut_down()
        except Exception:  # noqa: BLE001
            pass


class _HwmonSource:
    name = "sysfs hwmon"

    def __init__(self, device_index, bus_id=None):
        cards = []
        for d in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
            try:
                if open(os.path.join(d, "vendor")).read().strip() != "0x1002":
                    continue
            except OSError:
                continue
            hw = sorted(glob.glob(os.path.join(d, "hwmon", "hwmon*")))
            if hw:
                cards.append((os.path.realpath(d), hw[0]))
        if not cards:
            raise RuntimeError("no amdgpu hwmon directory")
        pick = None
        if bus_id:
            pick = next((c for c in cards if c[0].lower().endswith(bus_id.lower())), None)
        self.dev, self.hw = pick or cards[min(device_index, len(cards) - 1)]
        self.pfile = next((p for p in (os.path.join(self.hw, n) for n in ("power1_input", "power1_average")) if os.path.exists(p)), None)
        self.ffile = os.path.join(self.hw, "freq1_input")
        if self.pfile is None and not os.path.exists(self.ffile):
            raise RuntimeError("hwmon has neither power nor frequency files")
        self.sample()

    @staticmethod
    def _read(path, scale):
        try:
            return float(open(path).read().strip()) * scale
        except (OSError, ValueError):
            return None

    def sample(self):
        return dict(sclk=self._read(self.ffile, 1e-6), sclk_xcd=None, power=self._read(self.pfile, 1e-6) if self.pfile else None,
                    energy=None, fw_ts=None, temp=self._read(os.path.join(self.hw, "temp2_input"), 1e-3), mclk=None, activity=None,
                    throttle=None)

    def cap(self):
        return self._read(os.path.join(self.hw, "power1_cap"), 1e-6)

    def close(self):
        pass


class _RocmSmiSource:
    name = "rocm-smi --json (subprocess, slow)"

    def __init__(self, device_index, bus_id=None):
        self.key = f"card{device_index}"
        self.sample()

    def _call(self, *flags):
        out = subprocess.run(["rocm-smi", *flags, "--json"], capture_output=True, text=True, timeout=20).stdout
        d = json.loads(out[out.index("{"):])
        return d.get(self.key) or next(iter(d.values()))

    def sample(self):
        d = self._call("--showpower", "--showclocks")
        power = sclk = None
        for k, v in d.items():
            kl = k.lower()
            try:
                if "power" in kl and "(w)" in kl and power is None:
                    power = float(v)
                if kl.startswith("sclk clock speed") and sclk is None:
                    sclk = float(str(v).strip("()").lower().replace("mhz", ""))
            except ValueError:
                pass
        if power is None and sclk is None:
            raise RuntimeError("rocm-smi returned neither power nor sclk")
        return dict(sclk=sclk, sclk_xcd=None, power=power, energy=None, fw_ts=None, temp=None, mclk=None, activity=None, throttle=None)

    def cap(self):
        try:
            d = self._call("--showmaxpower")
            return next((float(v) for k, v in d.items() if "max" in k.lower() and "power" in k.lower()), None)
        except Exception:  # noqa: BLE001
            return None

    def close(self):
        pass


def open_source(device_index=0, bus_id=None):
    """First telemetry source that answers on this box, or (None, reasons)."""
    why = []
    for cls in (_AmdSmiSource, _HwmonSource, _RocmSmiSource):
        try:
            return cls(device_index, bus_id), why
        except Exception as e:  # noqa: BLE001
            why.append(f"{cls.name}: {type(e).__name__}: {e}")
    return None, why


def torch_bus_id(device_index=0):
    """'bb:dd.f' of torch's device (matches the tail of an amdsmi / sysfs BDF), or None."""
    try:
        import torch
        p = torch.cuda.get_device_properties(device_index)
        return f"{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
    except Exception:  # noqa: BLE001
        return None


class _Child:
    """The sampler as a child process: ``start`` / ``stop`` / ``cap`` over its stdin, one JSON line back per command."""

    def __init__(self, device_index, bus_id, hz):
        import subprocess
        import sys
        self.p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--serve", str(device_index), bus_id or "-", str(hz)],
                                  stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, bufsize=1)
        hello = self._read(20.0)
        if not hello or hello.get("source") is None:
            why = (hello or {}).get("error", "the sampler process did not answer")
            self.close()
            raise RuntimeError(why)
        self.name = hello["source"] + " (child process)"
        self._cap = hello.get("cap")

    def _read(self, timeout):
        import select
        r, _, _ = select.select([self.p.stdout], [], [], timeout)
        if not r:
            return None
        line = self.p.stdout.readline()
        try:
            return json.loads(line) if line else None
        except ValueError:
            return None

    def ask(self, cmd, timeout=5.0):
        try:
            self.p.stdin.write(cmd + "\n")
            self.p.stdin.flush()
        except (OSError, ValueError):
            return None
        return self._read(timeout)

    def cap(self):
        return self._cap

    def close(self):
        try:
            self.p.stdin.close()
            self.p.terminate()
        except Exception:  # noqa: BLE001
            pass


def _serve(argv):
    """``telemetry.py --serve <device> <bus id | -> <hz>``: the child's main loop."""
    import sys
    dev, bus, hz = int(argv[0]), (None if argv[1] == "-" else argv[1]), float(argv[2])
    src, why = open_source(dev, bus)
    out = sys.stdout
    out.write(json.dumps(dict(source=None if src is None else src.name, cap=None if src is None else src.cap(), error="; ".join(why))) + "\n")
    out.flush()
    if src is None:
        return
    t = Telemetry(dev, hz=hz, source=src)
    for line in sys.stdin:
        cmd = line.strip()
        if cmd == "start":
            t.start()
            out.write("{}\n")
        elif cmd in ("stop", "stop_raw"):
            t.stop()
            ans = dict(summary=t.summary(), t0=t.t0, t1=t.t1)
            if cmd == "stop_raw":
                ans["samples"] = t.samples
            out.write(json.dumps(ans) + "\n")
        elif cmd == "quit":
            break
        else:
            out.write("{}\n")
        out.flush()


class Telemetry:
    """Samples shader clock and socket power of one GPU between ``start()`` and ``stop()`` (child process by default; a side
    thread of this process when a ``source`` object is given or TFIMM_TELEMETRY=inproc)."""
    # firmware units of the MI300-class gpu_metrics table: energy_accumulator 15.259 uJ, firmware_timestamp 10 ns
    ENERGY_UJ = 15.259
    FW_TICK_S = 1e-8
    _shared = {}

    def __init__(self, device_index=0, hz=200.0, source=None, raw=False):
        self.period = 1.0 / hz
        self.child = None
        self.raw = raw                      # child mode: bring the samples over as well (tools/power_probe.py cuts its own windows)
        self._summary = None
        mode = os.environ.get("TFIMM_TELEMETRY", "proc")
        if source is not None:
            self.src, self.why = source, []
        elif mode == "off":
            self.src, self.why = None, ["TFIMM_TELEMETRY=off"]
        elif mode == "inproc":
            key = device_index
            if key not in Telemetry._shared:            # one handle per process and device: amdsmi_init is not cheap
                Telemetry._shared[key] = open_source(device_index, torch_bus_id(device_index))
            self.src, self.why = Telemetry._shared[key]
        else:
            key = ("child", device_index, hz)
            if key not in Telemetry._shared:
                try:
                    Telemetry._shared[key] = (_Child(device_index, torch_bus_id(device_index), hz), [])
                except Exception as e:  # noqa: BLE001
                    Telemetry._shared[key] = (None, [f"sampler process: {type(e).__name__}: {e}"])
            self.child, self.why = Telemetry._shared[key]
            self.src = self.child
        self.samples = []
        self._stop = threading.Event()
        self._thread = None
        self.t0 = self.t1 = None

    def _loop(self):
        nxt = time.perf_counter()
        while not self._stop.is_set():
            try:
                s = self.src.sample()
                s["t"] = time.perf_counter()
                self.samples.append(s)
            except Exception:  # noqa: BLE001
                pass
            nxt += self.period
            d = nxt - time.perf_counter()
            if d > 0:
                self._stop.wait(d)
            else:
                nxt = time.perf_counter()

    def start(self):
        self.samples = []
        self._summary = None
        self._stop.clear()
        self.t0 = time.perf_counter()
        if self.child is not None:
            if self.child.ask("start") is None:           # the sampler died: keep measuring without it
                self._drop_child()
            return self
        if self.src is not None:
            self._thread = threading.Thread(target=self._loop, name="tfimm-telemetry", daemon=True)
            self._thread.start()
        return self

    def _drop_child(self):
        """A sampler that missed an answer is closed and forgotten by EVERY instance: its late reply would otherwise be read as
        the answer to the next instance's command (the child is shared per device through ``_shared``)."""
        child = self.child
        for key in [k for k, v in Telemetry._shared.items() if v[0] is child]:
            del Telemetry._shared[key]
        if child is not None:
            child.close()
        self.why = ["the sampler process stopped answering"]
        self.child = self.src = None

    def stop(self):
        self.t1 = time.perf_counter()
        if self.child is not None:
            ans = self.child.ask("stop_raw" if self.raw else "stop", timeout=10.0)
            if not ans or "summary" not in ans:
                self._drop_child()
            else:
                self._summary = ans["summary"]
                self.samples = ans.get("samples", [])
            return self
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=5.0)
            self._thread = None
        return self

    __enter__ = start

    def __exit__(self, *exc):
        self.stop()
        return False

    def summary(self, digits=1):
        if self._summary is not None:
            return self._summary
        if self.src is None or self.child is not None:
            return dict(source=None, error="; ".join(self.why) or "no telemetry source")
        ss = self.samples
        out = dict(source=self.src.name, samples=len(ss), window_s=round((self.t1 or time.perf_counter()) - self.t0, 4))
        out["hz"] = round(len(ss) / max(out["window_s"], 1e-9), 1)

        def agg(key, name):
            v = [s[key] for s in ss if s.get(key) is not None]
            if v:
                out[f"{name}_mean"] = round(sum(v) / len(v), digits)
                out[f"{name}_min"] = round(min(v), digits)
                out[f"{name}_max"] = round(max(v), digits)
            else:
                out[f"{name}_mean"] = None
        agg("sclk", "sclk_mhz")
        agg("power", "power_w")
        agg("mclk", "mclk_mhz")
        agg("temp", "temp_hotspot_c")
        xcd = [s["sclk_xcd"] for s in ss if s.get("sclk_xcd")]
        if xcd:
            n = min(len(x) for x in xcd)
            out["sclk_mhz_mean_per_xcd"] = [round(sum(x[i] for x in xcd) / len(xcd), digits) for i in range(n)]
        # mean power from the firmware's energy accumulator over (first sample, last sample): independent of the sampling rate
        e = [(s["energy"], s["fw_ts"], s["t"]) for s in ss if s.get("energy") is not None]
        if len(e) >= 2 and e[-1][0] > e[0][0]:
            de = (e[-1][0] - e[0][0]) * self.ENERGY_UJ * 1e-6
            dt_fw = (e[-1][1] - e[0][1]) * self.FW_TICK_S if (e[0][1] is not None and e[-1][1] is not None) else 0.0
            dt = dt_fw if dt_fw > 0 else (e[-1][2] - e[0][2])
            if dt > 0:
                out["energy_power_w"] = round(de / dt, digits)
                out["energy_window_s"] = round(dt, 4)
        thr = [s["throttle"] for s in ss if s.get("throttle") is not None]
        if thr:
            out["throttle_status_or"] = hex(int(__import__("functools").reduce(lambda a, b: a | b, thr)))
        cap = self.src.cap()
        out["power_cap_w"] = None if cap is None else round(cap, digits)
        return out


if __name__ == "__main__":
    import sys
    if len(sys.argv) >= 5 and sys.argv[1] == "--serve":
        _serve(sys.argv[2:5])
