"""One small pool of side HIP streams per device, shared by everything in the package that overlaps launches
(the two-stream inference path, the training step's leaf / selector / coordinate-gradient streams, the gradient
all-reduce).

Why a pool: a ROCm process has a handful of hardware queues per device (4 unless GPU_MAX_HW_QUEUES says otherwise)
and streams are dealt onto them round-robin.  Streams created here and there -- two by the inference path, three by
the training step, one per FlatParamGroup -- soon share a queue with the stream they were meant to run beside, and
the overlap silently turns into serialisation: the training step measured 23.7 ms in a fresh process and 25.5 ms
after an inference pass of the same process had created its two streams.  With the pool the package never holds
more than POOL side streams per device, whatever ran before."""
import torch

POOL = 3
_streams = {}


def side(device, i):
    """The i-th shared side stream of ``device`` (i is taken modulo POOL)."""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    pool = _streams.get(idx)
    if pool is None:
        pool = _streams[idx] = [torch.cuda.Stream(device=torch.device('cuda', idx)) for _ in range(POOL)]
    return pool[i % POOL]
