"""One rank of the 2-GPU exchange test (launched by tests/test_two_rank_gpu.py through torch.distributed.run): each rank owns
half of a 200k x 384 corpus, searches it locally (row_base = its offset) and the per-shard top-10 lists are exchanged and
merged (a) by the C-ABI path -- rmu_shard_allgather_topk: ONE ncclAllGather issued from librmu.so -- and (b) by
torch.distributed's all_gather + rmu_topk_merge; both must equal the oracle's search of the whole corpus."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from ragmeup_amd import FlatIndex  # noqa: E402
from ragmeup_amd.shard import NativeComm, ShardedSearcher, shard_bounds  # noqa: E402
from tests.helpers import assert_topk_parity  # noqa: E402

rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
x = O.make_corpus(200_000)
q, planted = O.make_queries(x, 300)
lo, hi = shard_bounds(len(x), world, rank)
idx = FlatIndex(384, device=local)
idx.add(x[lo:hi])
qd = torch.from_numpy(q).cuda()
comm = NativeComm.from_torch_dist(device=local)
s1, r1 = ShardedSearcher(idx, row_base=lo, comm=comm).search(qd, 10)
s2, r2 = ShardedSearcher(idx, row_base=lo).search(qd, 10)
assert torch.equal(r1, r2) and torch.equal(s1, s2), "C-ABI exchange and torch.distributed exchange differ"
os_, or_ = O.flat_search(q, x, 14)
assert_topk_parity(s1.cpu().numpy(), r1.cpu().numpy(), os_, or_)
assert (r1[:, 0].cpu().numpy() == planted).all()
# every rank holds the same merged result
chk = r1.clone()
dist.broadcast(chk, 0)
assert torch.equal(chk, r1)
dist.barrier()
comm.close()
idx.close()
dist.destroy_process_group()
if rank == 0:
    print("TWO_RANK_OK")
