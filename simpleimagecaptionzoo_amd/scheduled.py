"""Scheduled sampling in the XE forward (BUTD_Model.py:120-132, AoA_Model.py:258-270, NIC_Model.py:77-89).

The reference's epoch driver raises `model.ss_prob` per epoch (Engine.py:140-144, on by default: Main.py:166-169) but sets
it on the Captioner while each DecoderRNN reads its own attribute, so the reference's runs never leave teacher forcing.
The drop-in default is therefore the same: the Captioner accepts the attribute and ignores it.  `scheduled_sampling =
True` on the Captioner makes the attribute live: the device path then does what DecoderRNN.forward does when its own
`ss_prob` is set (pinned by goldens generated from the reference decoders with exactly that attribute set)."""
import torch

from ._lib import check, lib, ptr


def handle_set_scheduled_sampling(handle, fn_name, ss_prob, gate=None, draw=None):
    """icz_<family>_set_scheduled_sampling on a handle wrapper; gate / draw: optional explicit uniforms [T, B]."""
    keep = [None if u is None else torch.as_tensor(u, dtype=torch.float32).to(handle.device).contiguous() for u in (gate, draw)]
    check(getattr(lib(), fn_name)(handle._h, float(ss_prob), ptr(keep[0]), ptr(keep[1])))
    handle._ss_live = keep           # the library reads them during the next xe_forward


def scheduled_sampling_prob(epoch, ss_opts):
    """The schedule of Engine.py:140-144: 0 until `ss_start_epoch` (or for a negative start), then `ss_inc_prob` more every
    `ss_inc_every` epochs, capped at `ss_max_prob`."""
    start = ss_opts["ss_start_epoch"]
    if start < 0 or epoch <= start:
        return 0.0
    return min(ss_opts["ss_inc_prob"] * ((epoch - start) // ss_opts["ss_inc_every"]), ss_opts["ss_max_prob"])


class ScheduledSamplingState:
    """Captioner side: the `ss_prob` attribute Engine.py:143 sets, the `scheduled_sampling` switch that makes it live,
    optional explicit draws (parity tests), and the push of the effective value to the device handle when it changed."""

    def _ss_init(self):
        self.ss_prob = 0.0
        self.scheduled_sampling = False          # False: `ss_prob` is ignored, as in the reference's own runs
        self._ss_draws = (None, None)
        self._ss_bound = None

    def set_scheduled_sampling_draws(self, gate=None, draw=None):
        """Explicit uniforms [T, B] for the gate and the draw of the next forward; None = the library's Philox streams."""
        self._ss_draws = (gate, draw)

    def _ss_push(self, handle, fresh_handle=False):
        off = (0.0, id(None), id(None))                     # a new handle starts switched off
        bound = off if (fresh_handle or self._ss_bound is None) else self._ss_bound
        prob = float(self.ss_prob) if self.scheduled_sampling else 0.0
        key = (prob, id(self._ss_draws[0]), id(self._ss_draws[1]))
        if key != bound:
            handle.set_scheduled_sampling(prob, *self._ss_draws)
        self._ss_bound = key
