"""Streaming video stabilisation: the device-side counterpart of the reference's ``process()`` loop
(reference main_new.py:612-741) minus its OpenCV I/O (decode, BGR->gray, INTER_AREA resize, MJPG encode stay with the
caller -- cv2 is not part of the accelerated path).

Per frame i the reference builds a 31-frame window of 256x256 gray frames [i-15, i+15] (the first / last frame
repeated at the ends, main_new.py:627-633,653-660), runs ``netG(window, False)``, resizes the 256x256 field to the frame
size (UpsamplingBilinear2d) and warps the full-resolution RGB frame (grid_sample), one frame at a time with a
host<->device round trip per frame.  Here:
  * frames are processed in batches of ``batch`` windows per generator call (the windows are overlapping views of one
    gray-frame buffer, gathered on the device);
  * field resize + warp is the fused HIP kernel (the resized field is never materialised);
  * when the inputs live in (pinned) host memory, batch k+1 is copied H2D on a side stream while batch k computes,
    and results go D2H on a third stream (hipMemcpyAsync through torch, one event per hand-off);
  * several GPUs take contiguous frame chunks with a 15-frame halo each (``distributed.shard_frames``): no collective.
"""
import torch

from . import functional as PF
from . import hipabi as A


def window_planes(frames_hwc, size=256, frames_are_rgb=False, out=None):
    """``cv2.resize(cv2.cvtColor(frame, COLOR_BGR2GRAY), (size, size), INTER_AREA) / 255 * 2 - 1`` for a batch of decoded
    frames (reference main_new.py:639-643,653-667) on the device: (T, H, W, 3) uint8 -> (T, size, size) float32."""
    A.require_cuda(frames_hwc, dtype=torch.uint8)
    if frames_hwc.dim() != 4 or frames_hwc.shape[3] != 3:
        raise ValueError("window_planes: frames must be (T, H, W, 3) uint8, got %s" % (tuple(frames_hwc.shape),))
    frames_hwc = frames_hwc.contiguous()
    t, h, w, _ = frames_hwc.shape
    if out is None:
        out = torch.empty((t, size, size), device=frames_hwc.device, dtype=torch.float32)
    A.require_cuda(out)
    if tuple(out.shape) != (t, size, size) or not out.is_contiguous():
        raise ValueError("window_planes: out must be a contiguous (T, %d, %d) float32 tensor" % (size, size))
    A.check(A.lib().pws_gray_area_u8(A.ptr(frames_hwc), A.ptr(out), t, h, w, size, size, 1, int(bool(frames_are_rgb)),
                                     A.current_stream()), "pws_gray_area_u8")
    return out


def area_half(frames_hwc, swap_rb=False):
    """``cv2.cvtColor(cv2.resize(frame, (W/2, H/2), INTER_AREA), COLOR_BGR2RGB)`` of the output frames (main_new.py:723-725;
    the swap is optional): (T, H, W, 3) uint8 -> (T, H/2, W/2, 3) uint8."""
    A.require_cuda(frames_hwc, dtype=torch.uint8)
    frames_hwc = frames_hwc.contiguous()
    t, h, w, _ = frames_hwc.shape
    out = torch.empty((t, h // 2, w // 2, 3), device=frames_hwc.device, dtype=torch.uint8)
    A.check(A.lib().pws_area_half_u8(A.ptr(frames_hwc), A.ptr(out), t, h, w, int(bool(swap_rb)), A.current_stream()),
            "pws_area_half_u8")
    return out


def area_resize(frames_hwc, size, swap_rb=False):
    """``cv2.resize(frame, size, interpolation=cv2.INTER_AREA)`` for any down-scaling ratio, ``size`` = (width, height) as cv2 takes
    it (main_new.py:723: ``cv2.resize(samples, (640, 360), interpolation=cv2.INTER_AREA)``): (T, H, W, 3) uint8 -> (T, height, width, 3)."""
    A.require_cuda(frames_hwc, dtype=torch.uint8)
    frames_hwc = frames_hwc.contiguous()
    t, h, w, _ = frames_hwc.shape
    ow, oh = int(size[0]), int(size[1])
    out = torch.empty((t, oh, ow, 3), device=frames_hwc.device, dtype=torch.uint8)
    A.check(A.lib().pws_area_resize_u8(A.ptr(frames_hwc), A.ptr(out), t, h, w, oh, ow, int(bool(swap_rb)), A.current_stream()),
            "pws_area_resize_u8")
    return out


def _windows(gray_padded, start, count, period):
    """(count, period+1, 256, 256) windows: window b covers padded frames [start+b, start+b+period].  An OVERLAPPING VIEW of the
    plane buffer (sample stride = one plane): the generator's first layer reads it in place (pws_netg_opts.x_sample_stride); no
    31-plane copy per window (65 MB per batch of 8)."""
    h, w = gray_padded.shape[-2:]
    return torch.as_strided(gray_padded, (count, period + 1, h, w), (h * w, h * w, w, 1),
                            storage_offset=gray_padded.storage_offset() + start * h * w)


class VideoStabilizer:
    def __init__(self, netG, batch=32, period=30, device=None, swap_rb=False):
        self.net = netG
        self.swap_rb = bool(swap_rb)   # uint8 frames only: BGR (cv2) in, RGB out, as main_new.py:679
        self.batch = int(batch)
        self.period = int(period)
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        # grid_sample follows the generator's coordinate convention (UnetGenerator.align_corners: torch >= 1.3 default False;
        # True for checkpoints trained under the reference's pinned torch 0.4)
        self.align_corners = bool(getattr(getattr(netG, "module", netG), "align_corners", False))
        self._lanes = []   # compute streams of run_video(in_flight > 1)

    def _pad(self, gray, halo_left, halo_right):
        """Replicates the first / last available frame so that every frame owns a full window (reference :627-633)."""
        half = self.period // 2
        need_l, need_r = half - halo_left, half - halo_right
        parts = []
        if need_l > 0:
            parts.append(gray[:1].expand(need_l, -1, -1))
        parts.append(gray)
        if need_r > 0:
            parts.append(gray[-1:].expand(need_r, -1, -1))
        return torch.cat(parts, 0) if len(parts) > 1 else gray

    @torch.no_grad()
    def run(self, gray, frames, halo_left=0, halo_right=0, out=None):
        """gray: (T + halo_left + halo_right, 256, 256) float32 in [-1,1] (``x/255*2-1`` as the reference, :650) for the T
        frames to stabilise plus whatever real neighbours exist on either side (up to period//2); frames: (T, C, H, W)
        float32 0..255, or (T, H, W, 3) uint8 as OpenCV delivers them (then the warp runs in the uint8 kernel: 4x less
        PCIe and HBM traffic per frame, ``self.swap_rb`` applies the reference's BGR->RGB).  Tensors may be on the device or in
        (pinned) host memory.  Returns a tensor of the frames' shape and dtype on the inputs' side."""
        half = self.period // 2
        if not (0 <= halo_left <= half and 0 <= halo_right <= half):
            raise ValueError("halo must be within [0, %d]" % half)
        T = frames.shape[0]
        if gray.shape[0] != T + halo_left + halo_right:
            raise ValueError("gray has %d frames, expected %d + %d + %d" % (gray.shape[0], T, halo_left, halo_right))
        on_host = not frames.is_cuda
        dev = self.device
        gray_d = gray.to(dev, non_blocking=True) if not gray.is_cuda else gray  # 256 KB per frame: copied once
        gp = self._pad(gray_d.contiguous(), halo_left, halo_right).contiguous()
        if out is None:
            out = torch.empty_like(frames, pin_memory=on_host) if on_host else torch.empty_like(frames)
        if T == 0:
            return out
        compute = torch.cuda.current_stream(dev)
        h2d, d2h = (torch.cuda.Stream(dev), torch.cuda.Stream(dev)) if on_host else (None, None)

        def fetch(s):
            e = min(T, s + self.batch)
            if not on_host:
                return frames[s:e], None
            with torch.cuda.stream(h2d):
                buf = frames[s:e].to(dev, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(h2d)
            return buf, ev

        nxt = fetch(0)
        pending = []
        for s in range(0, T, self.batch):
            e = min(T, s + self.batch)
            cur, ev = nxt
            if e < T:
                nxt = fetch(e)  # overlaps with this batch's compute
            if ev is not None:
                compute.wait_event(ev)
            win = _windows(gp, s, e - s, self.period)
            field = self.net(win, False)
            if cur.dtype == torch.uint8:
                warped = PF.upsample_grid_sample_u8(cur.contiguous(), field, swap_rb=self.swap_rb, align_corners=self.align_corners)
            else:
                warped = PF.upsample_grid_sample(cur.contiguous(), field, align_corners=self.align_corners)
            if on_host:
                done = torch.cuda.Event()
                done.record(compute)
                with torch.cuda.stream(d2h):
                    d2h.wait_event(done)
                    out[s:e].copy_(warped, non_blocking=True)
                warped.record_stream(d2h)
                cur.record_stream(compute)
                pending.append(warped)
            else:
                out[s:e].copy_(warped)
        if on_host:
            d2h.synchronize()
        return out

    @torch.no_grad()
    def run_video(self, frames, chunk=32, half_size_output=False, frames_are_rgb=False, halo_left=0, halo_right=0, crop=None,
                  output_size=None, in_flight=1):
        """The whole device side of the reference's ``process()`` loop for one decoded clip: ``frames`` (T, H, W, 3) uint8 as
        cv2 delivers them, in (pinned) host memory or on the device.  Per chunk of ``chunk`` frames: H2D on a side stream,
        gray + INTER_AREA 256x256 window planes computed there from the uploaded frames (so nothing but the uint8 frames
        crosses PCIe and the host does no per-frame image processing), the generator on batched windows, the fused uint8
        resize+warp, optionally the 2x INTER_AREA down-scale of the output (main_new.py:723), D2H on a third stream.  Chunk
        k+1 uploads while chunk k computes; window planes are kept for the whole clip (256 KB per frame), frames per chunk.
        halo_left / halo_right: ``frames`` additionally holds that many real neighbour frames before / after the T frames to
        stabilise (a rank's shard of a longer clip, ``distributed.shard_frames``: up to period//2 each); they only feed the
        windows.  crop = (x_start, x_end, y_start, y_end[, threshold]): the reference's crop window of the OUTPUT frame
        (main_new.py:607-610,729-730: ``samples[int(y_start)+threshold:int(y_end)-threshold, int(x_start)+threshold:
        int(x_end)-threshold]``, applied after the down-scale), sliced on the device so that only the cropped frames cross PCIe;
        the reference's live values (0, 640, 0, 360, threshold 0) are the whole 640x360 frame.  The 3x3 GaussianBlur(sigma=0.2)
        that follows in the reference (:731) is the identity on 8-bit frames and is not run.
        output_size = (width, height): the reference's ``cv2.resize(samples, (640, 360), INTER_AREA)`` for ANY source size
        (main_new.py:723; half_size_output=True is the same thing for a 1280 x 720 source).
        in_flight = 2: consecutive chunks compute on two streams without waiting for each other (one chunk walks the generator's chain of
        small levels while the other's large layers fill the chip: DESIGN.md section 8); for the duration of the call the generator runs one
        queue per forward and, in graph mode, keeps one graph + arena per stream.  Same frames bit for bit as in_flight = 1.
        Returns (T, H, W, 3) -- or (T, H/2, W/2, 3) / (T, height, width, 3), or the crop window of it -- uint8 on the inputs' side."""
        half = self.period // 2
        if not (0 <= halo_left <= half and 0 <= halo_right <= half):
            raise ValueError("halo must be within [0, %d]" % half)
        n_all = frames.shape[0]
        T = n_all - halo_left - halo_right
        if T < 0:
            raise ValueError("frames holds %d frames, fewer than the halos %d + %d" % (n_all, halo_left, halo_right))
        on_host = not frames.is_cuda
        dev = self.device
        h, w = frames.shape[1], frames.shape[2]
        if output_size is not None and half_size_output:
            raise ValueError("run_video: give output_size or half_size_output, not both")
        if output_size is not None:
            ow, oh = int(output_size[0]), int(output_size[1])
            if not (0 < ow <= w and 0 < oh <= h):
                raise ValueError("run_video: output_size %s must not exceed the %d x %d source (INTER_AREA down-scaling)" % (tuple(output_size), w, h))
        else:
            oh, ow = (h // 2, w // 2) if half_size_output else (h, w)
        cy0, cy1, cx0, cx1 = 0, oh, 0, ow
        if crop is not None:
            if len(crop) not in (4, 5):
                raise ValueError("crop must be (x_start, x_end, y_start, y_end[, threshold])")
            th = int(crop[4]) if len(crop) == 5 else 0
            cx0, cx1, cy0, cy1 = int(crop[0]) + th, int(crop[1]) - th, int(crop[2]) + th, int(crop[3]) - th
            if not (0 <= cx0 < cx1 <= ow and 0 <= cy0 < cy1 <= oh):
                raise ValueError("crop window x [%d, %d) y [%d, %d) does not lie inside the %d x %d output frame" % (cx0, cx1, cy0, cy1, ow, oh))
        oshape = (T, cy1 - cy0, cx1 - cx0, 3)
        out = torch.empty(oshape, dtype=torch.uint8, pin_memory=True) if on_host else torch.empty(oshape, dtype=torch.uint8, device=dev)
        if T == 0:
            return out
        compute = torch.cuda.current_stream(dev)
        up, down = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        in_flight = max(1, int(in_flight))
        # (the lanes live as long as this object: the generator keeps an arena / a graph per STREAM, and torch's allocator a pool per stream)
        # The caller's stream is lane 0: the runtime deals streams to a handful of hardware queues, and a lane that shares its queue with
        # the upload or the download stream waits behind their copies (two NEW lanes beside up / down: 1 744 -> 1 291 frames/s).
        while len(self._lanes) < in_flight - 1:
            self._lanes.append(torch.cuda.Stream(dev))
        lanes = [compute] + self._lanes[:in_flight - 1]
        mod = getattr(self.net, "module", self.net)
        restore = None
        if in_flight > 1:
            restore = (mod._graph_mode, mod._graph_alias, getattr(mod, "_graph_per_stream", False), mod.two_queues)
            mod.two_queues = False   # two forwards that both fork into the device's one side queue would serialise there
            if mod._graph_mode:
                mod.enable_graph(True, mod._graph_alias, per_stream=True)
            for lane in lanes[1:]:
                lane.wait_stream(compute)
        gray_all = torch.empty((n_all, 256, 256), device=dev, dtype=torch.float32)
        # chunk bounds in the index space of `frames`: the first / last chunk carry the halo frames with them
        chunks = [(s if s > halo_left else 0, min(halo_left + T, s + chunk) if s + chunk < halo_left + T else n_all)
                  for s in range(halo_left, halo_left + T, chunk)]
        up.wait_stream(compute)   # gray_all was allocated on the compute stream

        def upload(c):
            s, e = chunks[c]
            with torch.cuda.stream(up):
                buf = frames[s:e].to(dev, non_blocking=True) if on_host else frames[s:e]
                window_planes(buf, 256, frames_are_rgb, out=gray_all[s:e])
                ev = torch.cuda.Event()
                ev.record(up)
            return buf, ev
        staged, uploaded = {}, 0
        try:
            for c, (s, e) in enumerate(chunks):
                lane = lanes[c % len(lanes)]
                # a chunk's windows reach `half` frames past its end: every chunk that holds one of those frames must be on the
                # device (and its planes computed) first; one chunk further ahead keeps the upload stream busy during this compute
                need = min(n_all, e + half)
                while uploaded < len(chunks) and (chunks[uploaded][0] < need or uploaded <= c + 1):
                    staged[uploaded] = upload(uploaded)
                    uploaded += 1
                for k in range(c, uploaded):
                    if chunks[k][0] < need:
                        lane.wait_event(staged[k][1])
                # (planes of EARLIER chunks that this one's windows reach back into were produced before staged[c]'s event on the same stream)
                buf, ev = staged.pop(c)
                # core frames of this chunk (the halo frames at the clip's ends are uploaded with it but not stabilised)
                cs, ce = max(s, halo_left), min(e, halo_left + T)
                hl, hr = min(half, cs), min(half, n_all - ce)
                with torch.cuda.stream(lane):
                    warped = self.run(gray_all[cs - hl:ce + hr], buf[cs - s:ce - s], halo_left=hl, halo_right=hr)
                    if half_size_output:
                        warped = area_half(warped)
                    elif output_size is not None and (oh, ow) != (h, w):
                        warped = area_resize(warped, (ow, oh))
                    if crop is not None:
                        warped = warped[:, cy0:cy1, cx0:cx1, :].contiguous()   # packed on the device: only the window crosses PCIe
                    s, e = cs - halo_left, ce - halo_left   # position in the output
                    if on_host:
                        buf.record_stream(lane)
                        done = torch.cuda.Event()
                        done.record(lane)
                        with torch.cuda.stream(down):
                            down.wait_event(done)
                            out[s:e].copy_(warped, non_blocking=True)
                        warped.record_stream(down)
                    else:
                        out[s:e].copy_(warped)
        finally:
            if restore is not None:
                for lane in lanes[1:]:
                    compute.wait_stream(lane)   # the caller's stream sees every chunk; gray_all and the outputs may be freed behind it
                mod.two_queues = restore[3]
                mod.enable_graph(restore[0], restore[1], per_stream=restore[2])
        if on_host:
            down.synchronize()
        return out
