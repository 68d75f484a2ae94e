"""Input staging of the training loops on MI355X.

Reference (main_nce.py:207-209,299-302,310; main_coclr.py:221-223,366-368):

    transform_train_cuda = Compose([T.Normalize(mean, std, channel=1)])
    def tr(x):  return transforms_cuda(x).view(B,3,num_seq,seq_len,H,W).transpose(1,2).contiguous()
    input_seq = tr(input_seq.cuda(non_blocking=True))

i.e. fp32 frames over PCIe (403 MB/step at B=32), a normalise pass, a transposing copy, and two
more `.contiguous()` copies inside the model (model/pretrain.py:149-150).  `tr()` below does the
normalisation and the re-layout in ONE kernel (csrc/staging.hip) and also accepts the loader's
frames as uint8 (what they are before ToTensor: 4x less PCIe traffic), bit-identical to
`ToTensor` + `Normalize` on the same bytes.  The model already consumes `block[:, i]` as strided
views, so nothing else is copied.
"""
import torch

from . import ops

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def tr(x, num_seq, seq_len, mean=IMAGENET_MEAN, std=IMAGENET_STD, out=None):
    """x: (B, 3, num_seq*seq_len, H, W) on the device, fp32 in [0,1] (the loader's ToTensor output)
    or uint8 (raw frames) -> (B, num_seq, 3, seq_len, H, W) fp32, normalised per channel."""
    if x.dim() != 5 or x.shape[2] != num_seq * seq_len:
        raise ValueError("coclr_amd: expected frames of shape (B, C, %d, H, W), got %s"
                         % (num_seq * seq_len, tuple(x.shape)))
    if x.dtype not in (torch.uint8, torch.float32):
        raise TypeError("coclr_amd: frames must be uint8 or float32, got %s" % x.dtype)
    x = x.contiguous()
    B, C, _, H, W = x.shape
    if out is None:
        out = torch.empty(B, num_seq, C, seq_len, H, W, dtype=torch.float32, device=x.device)
    ops.stage_clips(x, out, num_seq, mean, std)
    return out


class ClipStager:
    """Host side of the uint8 path: two pinned buffers and a copy stream, so the H2D transfer of
    batch i+1 (101 MB at B=32 instead of 403 MB) overlaps the compute of batch i.

        stager = ClipStager(num_seq=2, seq_len=32)
        for frames_u8, ... in loader:                 # (B, 3, 64, 128, 128) uint8, host
            block = stager(frames_u8)                 # (B, 2, 3, 32, 128, 128) fp32, device
    """

    def __init__(self, num_seq, seq_len, mean=IMAGENET_MEAN, std=IMAGENET_STD, device=None):
        self.num_seq, self.seq_len, self.mean, self.std = num_seq, seq_len, mean, std
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None \
            else torch.device(device)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.pinned = [None, None]
        self.staged = [None, None]
        self.done = [None, None]
        self.read_done = [None, None]      # main-stream event: the kernel that last READ staged[f]
        self.flip = 0

    def __call__(self, frames):
        f = self.flip
        self.flip = 1 - f
        if frames.is_cuda:
            return tr(frames, self.num_seq, self.seq_len, self.mean, self.std)
        if self.done[f] is not None:
            self.done[f].synchronize()         # the transfer that last used this pinned buffer
        if not frames.is_pinned():
            if self.pinned[f] is None or self.pinned[f].shape != frames.shape or \
                    self.pinned[f].dtype != frames.dtype:
                self.pinned[f] = torch.empty(frames.shape, dtype=frames.dtype).pin_memory()
            self.pinned[f].copy_(frames)
            frames = self.pinned[f]
        main = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(self.copy_stream):
            if self.staged[f] is None or self.staged[f].shape != frames.shape or \
                    self.staged[f].dtype != frames.dtype:
                self.staged[f] = torch.empty(frames.shape, dtype=frames.dtype, device=self.device)
            elif self.read_done[f] is not None:
                # only the kernel that last read THIS buffer (two calls ago), not everything the
                # main stream has queued since: the transfer of batch i+1 runs under step i
                self.copy_stream.wait_event(self.read_done[f])
            self.staged[f].copy_(frames, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self.done[f] = ev
        main.wait_stream(self.copy_stream)
        out = tr(self.staged[f], self.num_seq, self.seq_len, self.mean, self.std)
        ev = torch.cuda.Event()
        ev.record(main)
        self.read_done[f] = ev
        return out
