"""Shader clock and socket power of one GPU, sampled by a background thread while a timed region runs (amdsmi; every
failure turns the sampler into a no-op: `summary()` is None).  Why bench.py carries it: the training step keeps the whole
chip busy for tens of milliseconds per step, and an MI355X then sits at its socket power cap with the shader clock well under
the 2.4 GHz the dense-MFMA peak (2.5 PFLOP/s bf16) is quoted at - the roofline fraction has to be read against that."""
import threading
import time


class ClockSampler:
    def __init__(self, pci_bus_id=None, period_s=0.02):
        self.period = period_s
        self.samples = []
        self._stop = threading.Event()
        self._thr = None
        self._h = None
        self._smi = None
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            hs = amdsmi.amdsmi_get_processor_handles()
            pick = None
            if pci_bus_id is not None:
                for h in hs:
                    try:
                        if amdsmi.amdsmi_get_gpu_device_bdf(h).lower().endswith(pci_bus_id.lower()):
                            pick = h
                            break
                    except Exception:
                        pass
            self._h = pick if pick is not None else (hs[0] if len(hs) == 1 else None)
            self._smi = amdsmi
            if self._h is not None:
                self._read()     # fail here rather than in the thread
        except Exception:
            self._h = None

    def _read(self):
        a = self._smi
        clk = a.amdsmi_get_clock_info(self._h, a.AmdSmiClkType.GFX)
        pw = a.amdsmi_get_power_info(self._h)
        mhz = clk.get("clk", clk.get("cur_clk"))
        w = pw.get("current_socket_power", pw.get("average_socket_power"))
        cap = pw.get("power_limit")
        return float(mhz), float(w), (float(cap) if isinstance(cap, (int, float)) else None)

    def _run(self):
        while not self._stop.is_set():
            try:
                self.samples.append(self._read())
            except Exception:
                pass
            time.sleep(self.period)

    def start(self):
        if self._h is None:
            return self
        self.samples = []
        self._stop.clear()
        self._thr = threading.Thread(target=self._run, daemon=True)
        self._thr.start()
        return self

    def stop(self):
        if self._thr is not None:
            self._stop.set()
            self._thr.join(timeout=2.0)
            self._thr = None
        return self.summary()

    def summary(self):
        if not self.samples:
            return None
        # drop the first fifth: the clock is still ramping from idle when the region starts
        s = self.samples[len(self.samples) // 5:] or self.samples
        mhz = [x[0] for x in s if x[0] > 0]
        w = [x[1] for x in s if x[1] > 0]
        cap = next((x[2] for x in s if x[2]), None)
        if not mhz:
            return None
        out = {"sclk_mhz_mean": round(sum(mhz) / len(mhz), 1), "sclk_mhz_min": round(min(mhz), 1), "sclk_mhz_max": round(max(mhz), 1),
               "socket_power_w_mean": round(sum(w) / len(w), 1) if w else None, "samples": len(s)}
        if cap:
            # amdsmi reports the limit in W on some versions and in uW on others
            out["socket_power_cap_w"] = round(cap / 1e6, 1) if cap > 1e5 else round(cap, 1)
        return out


def device_bus_id(index=0):
    """PCI address (bus:device.function tail) of torch's CUDA device `index`, for matching the amdsmi handle."""
    try:
        import torch
        p = torch.cuda.get_device_properties(index)
        return f"{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
    except Exception:
        return None
