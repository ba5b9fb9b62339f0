"""Cache of static launch plans (SipMaskEngine) that can never serve stale weights.

A launch plan snapshots the weights (BN folded, re-laid out as GEMM operands) at build time, so a cache keyed by
geometry alone would keep evaluating the weights of the first build after ``optimizer.step()`` or a checkpoint load
(mmcv's load_checkpoint goes through ``_load_from_state_dict``, not ``Module.load_state_dict``).  Every lookup
therefore compares a fingerprint of the live tensors -- (data_ptr, autograd version counter) of every parameter and
buffer -- with the one the cached plans were built from; any in-place update bumps a version counter
(``param.copy_``, ``optimizer.step``, and HipSGD bumps it explicitly after its raw-pointer update) and drops them.
A small LRU keeps the plans of the last few geometries (keep_ratio resizing yields many padded sizes in a real
evaluation run).

What the fingerprint cannot see: writes that bypass the version counter while keeping the storage -- ``p.data.copy_()``,
``p.data.mul_()`` (EMA hooks, hand-written optimizers), or a raw-pointer kernel on ``p.data``.  ``Module.load_state_dict``
therefore also drops the plans explicitly (``attach_invalidation``: a load_state_dict post-hook), ``hip_ops.sgd_step`` and
``HipSGD`` bump the counter themselves, and anything else of that kind must call ``invalidate()`` (``SipMask.invalidate_plans``).
``SIPMASK_PLAN_CHECKSUM=1`` adds a content check for debugging: a device-side sum of every tensor is compared on each
lookup (one sync per lookup -- not for production).
"""
import os
from collections import OrderedDict


_CHECKSUM = os.environ.get("SIPMASK_PLAN_CHECKSUM", "0") == "1"


def weights_version(tensors):
    """fingerprint of a set of live tensors: changes whenever one of them is updated in place or replaced"""
    ver = tuple((t.data_ptr(), t._version) for t in tensors)
    if _CHECKSUM:           # debug mode: also catches writes through .data / raw pointers (costs a device sync)
        import torch
        with torch.no_grad():
            ver += (float(sum(t.detach().double().sum() for t in tensors if t.is_floating_point())),)
    return ver


class PlanCache:
    def __init__(self, capacity=4):
        self.capacity = capacity
        self._plans = OrderedDict()
        self._version = None

    def get(self, key, tensors, build):
        """plan for ``key`` built from the CURRENT values of ``tensors`` (``build()`` is called on a miss)"""
        ver = weights_version(tensors)
        if ver != self._version:
            self._plans.clear()
            self._version = ver
        plan = self._plans.get(key)
        if plan is None:
            plan = build()
            self._plans[key] = plan
            while len(self._plans) > self.capacity:
                self._plans.popitem(last=False)
        else:
            self._plans.move_to_end(key)
        return plan

    def clear(self):
        self._plans.clear()
        self._version = None

    invalidate = clear      # public name: call after writing weights through .data / raw pointers

    def attach_invalidation(self, module):
        """drop the plans whenever `module.load_state_dict` has run (belt and braces beside the version fingerprint)"""
        module.register_load_state_dict_post_hook(lambda mod, incompatible: self.clear())
        return self

    def values(self):
        return self._plans.values()

    def __len__(self):
        return len(self._plans)


def module_tensors(module):
    return list(module.parameters()) + list(module.buffers())
