"""MI355X-native `DeepSVC` P-frame codec (reference: video_model.py:13-191).

Drop-in for the reference class: same constructor, attribute paths (=> same state_dict keys), and the
same `forward / forward1 / compress / decompress / update / load_state_dict / aux_loss*` signatures and
return values, so `test_video.py:230-236,342,349` drives it unchanged.  Underneath, one P-frame is a chain
of fused HIP launches over NHWC slices; see DESIGN.md section 3 for the dataflow and buffer plan.
"""
import math
from concurrent.futures import ThreadPoolExecutor

import torch
import torch.nn as nn

from . import _native as N
from . import ops
from .entropy_models import EntropyBottleneck
from .image_model import ChannelSplitICIP2020ResB, rename_legacy_keys
from .modules import (ME_Spynet, RefineNet, Reconstruction, FeatureExtraction, InterLayerPrediction)
from .ops import Act, to_act

F32 = torch.float32
# The motion decode chain on its own high-priority stream, overlapping the encoder's tail when compress() / decompress() are called back to
# back (test_video.py:342,349).  OPT-IN: `model.single_stream = True` on an instance, or DSVC_MV_STREAM=1 for all (DSVC_NO_MV_STREAM=1: none).
# [r5] In a process that holds ONE model (three streams: main, residual side stream, this one) it is worth 35.3 -> 34.85 ms per 1080p frame in
# f16 and **168.4 -> 163.5 ms in fp32** (whose 60 ms of encoder tail hide the whole chain; two alternating `bench.py --lanes 1` runs each).  It is
# not the default because the gain depends on how HIP maps streams onto the process's 4 hardware queues: with four GOP lanes it costs 7-9 %,
# four IDLE extra streams alone cost the 4-lane leg 3-4 %, and the same one-lane leg run inside bench.py's default process (8+ streams alive)
# came out SLOWER with it (fp32 5.71 vs 6.03 fps).  A full GPU test run with it on everywhere passed (287 tests): it is safe, not predictable.
_env = __import__("os").environ
MV_STREAM = [True if _env.get("DSVC_MV_STREAM", "") != "" else (False if _env.get("DSVC_NO_MV_STREAM", "") != "" else None)]     # None: per instance
HOST_TRACE = None          # tools/host_profile_1080p.py sets a list: (label, perf_counter) stamps of the host thread inside compress / decompress


def _stamp(label):
    if HOST_TRACE is not None:
        import time
        HOST_TRACE.append((label, time.perf_counter()))


_DECODE_POOL = ThreadPoolExecutor(max_workers=6, thread_name_prefix="dsvc-resdec")   # one per concurrent GOP lane (bench.py --lanes)


class DeepSVC(nn.Module):
    # True: decompress() runs the motion decode chain on its own stream beside the encoder's tail (opt-in: see MV_STREAM above)
    single_stream = False
    # True (one GOP in flight): the residual decode chain runs on a side stream beside the prediction segment; the multi-lane harness sets it
    # False on its lane models (one stream per lane: see decompress)
    use_side_stream = True

    def __init__(self):
        super().__init__()
        self.opticFlow = ME_Spynet()
        self.mv_codec = ChannelSplitICIP2020ResB(8, 64, 2)
        self.res_codec = ChannelSplitICIP2020ResB(64 + 6, 96, 64)
        self.MC = InterLayerPrediction()
        self.RefineMvNet = RefineNet(5, 64, 2)
        self.RefineResiNet = RefineNet(64 + 3, 64, 64)
        self.FeatureExtractor = FeatureExtraction(3, 64)
        self.enhance = Reconstruction(32 + 64, 64, 3, return_fea=True)

    # ------------------------------------------------------------------------------------------
    # stages shared by forward / compress / decompress
    # ------------------------------------------------------------------------------------------
    def _feature_in(self, feature):
        return None if feature is None else to_act(feature, ops.RT.storage)

    def _predict(self, ref, mv_coarse, sm_fea, feature):
        """mv refine -> warp -> inter-layer prediction (video_model.py:142-144 / 155-158)."""
        mv = self.RefineMvNet.run(mv_coarse, ref)
        warped = ops.warp(ref, mv)
        warp_fea, predict = self.MC.run(ref, warped, mv, sm_fea, feature)
        return warped, warp_fea, predict

    def _reconstruct(self, ref, predict_fea, res_coarse, warp_fea):
        """residual refine -> + prediction features -> enhancement (video_model.py:161-165).
        `predict_fea + recon_res` rides on the refine net's last conv as a second residual."""
        fea = self.RefineResiNet.run(res_coarse, ref, extra_residual=predict_fea)
        return self.enhance.run([fea, warp_fea])

    def graphs_ready(self, B, H, W, device):
        """True once every scan segment of both codecs has a captured HIP graph for frames of (padded) size HxW --
        from then on compress()/decompress() never capture again unless weights are repacked."""
        if not ops.USE_GRAPHS[0]:
            return True
        return all(codec.scan_graphs_ready(B, H // 16, W // 16, device) for codec in (self.mv_codec, self.res_codec))

    def _maybe_capture_graphs(self, ref):
        """HIP-graph capture of the codecs' scan segments, done here (main thread, no coder/decoder thread in flight)
        once the weight-packing epoch has been stable for a whole call -- i.e. after the first frame."""
        if not ops.USE_GRAPHS[0]:
            return
        epochs = lambda: (self.mv_codec._epoch_ref()[0], self.res_codec._epoch_ref()[0])
        epoch = epochs()
        if self.__dict__.get("_seen_epoch") != epoch:
            self.__dict__["_seen_epoch"] = epoch
            return
        for codec in (self.mv_codec, self.res_codec):
            codec.capture_scan_graphs(ref.B, ref.H // 16, ref.W // 16, ref.device)
        self.__dict__["_seen_epoch"] = epochs()

    @torch.no_grad()
    def warm_up(self, ref_frame, frames, sm_feas, max_frames=8):
        """Deterministic warm-up of THIS model instance on the calling thread's current stream (the caller makes sure no other thread
        is issuing HIP work: weight packing and graph capture want a quiescent device): codes chained P-frames -- frames[0] with
        feature = None (first-frame variant), the following ones in steady state, cycling through `frames` / `sm_feas` -- until every scan
        segment of both codecs has its HIP graph and the frame segments of both variants have been replayed once.  Everything a
        recording depends on belongs to this instance (per-codec packing epoch, per-model program cache), so other instances (GOP lanes)
        warming up before or after do not invalidate it.  -> number of frames coded; raises if the model is not ready after max_frames."""
        if len(frames) == 0 or len(sm_feas) == 0:
            raise ValueError("DeepSVC.warm_up: needs at least one P-frame and one semantic feature to code (got a lone key frame?)")
        ref, feat = ref_frame, None
        first = ops.to_act(ref_frame)
        B, H, W = first.B, first.H, first.W
        n = 0
        while n < max_frames:
            cur, sm = frames[n % len(frames)], sm_feas[n % len(sm_feas)]
            mv, res = self.compress(ref, cur, sm, feat)
            feat, ref, _, _ = self.decompress(ref, mv, res, sm, feat)
            n += 1
            if n >= 3 and self.graphs_ready(B, H, W, first.device):        # frame 1: first variant recorded; 2: steady recorded; 3: both replayed, graphs in place
                torch.cuda.current_stream(first.device).synchronize()
                return n
        raise RuntimeError(f"DeepSVC.warm_up: scan graphs not in place after {max_frames} frames (is another thread issuing HIP work / repacking weights?)")

    @staticmethod
    def _bpp(liks, pixels):
        total = sum(ops.reduce("sumlog", l) for l in liks)
        return (total / (-math.log(2) * pixels)).to(torch.float32)

    # ------------------------------------------------------------------------------------------
    # estimate path
    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, ref_frame, curr_frame, sm_fea, feature=None):
        """video_model.py:27-71 -> (recon, feature, mse, warp_mse, mc_mse, bpp_res, bpp_mv, bpp)."""
        ref, cur = to_act(ref_frame), to_act(curr_frame)
        pixels = cur.B * cur.H * cur.W
        mv_est = self.opticFlow.run(cur, ref)
        mv_coarse, y_lik, z_lik = self.mv_codec.run_forward([cur, mv_est, ref], x_hat_dtype=F32)
        warped, warp_fea, predict = self._predict(ref, mv_coarse, sm_fea, self._feature_in(feature))
        bpp_mv = self._bpp([y_lik, z_lik], pixels)
        predict_fea = self.FeatureExtractor.run(predict)
        cur_fea = self.FeatureExtractor.run(cur)
        res = ops.eltwise(N.EW_SUB, cur_fea, predict_fea)
        res_coarse, y_lik, z_lik = self.res_codec.run_forward([ref, res, predict])
        bpp_res = self._bpp([y_lik, z_lik], pixels)
        feat, recon = self._reconstruct(ref, predict_fea, res_coarse, warp_fea)
        n = float(cur.npix * cur.C)
        mse = (ops.reduce("sqdiff", recon, cur) / n).to(torch.float32)
        warp_loss = (ops.reduce("sqdiff", warped, cur) / n).to(torch.float32)
        mc_loss = (ops.reduce("sqdiff", predict, cur) / n).to(torch.float32)
        return recon.nchw(), feat.nchw(), mse, warp_loss, mc_loss, bpp_res, bpp_mv, bpp_mv + bpp_res

    @torch.no_grad()
    def forward1(self, ref_frame, curr_frame, sm_fea, feature=None):
        """video_model.py:73-94 -> (predict, warp_mse, mc_mse, bpp_mv)."""
        ref, cur = to_act(ref_frame), to_act(curr_frame)
        pixels = cur.B * cur.H * cur.W
        mv_est = self.opticFlow.run(cur, ref)
        mv_coarse, y_lik, z_lik = self.mv_codec.run_forward([cur, mv_est, ref], x_hat_dtype=F32)
        warped, _, predict = self._predict(ref, mv_coarse, sm_fea, self._feature_in(feature))
        n = float(cur.npix * cur.C)
        return (predict.nchw(), (ops.reduce("sqdiff", warped, cur) / n).to(torch.float32),
                (ops.reduce("sqdiff", predict, cur) / n).to(torch.float32), self._bpp([y_lik, z_lik], pixels))

    def _estimate(self, ref_frame, curr_frame, sm_fea, feature):
        """The estimate graph shared by forward / forward_msssim (video_model.py:27-71 == 96-135 up to the distortion)."""
        ref, cur = to_act(ref_frame), to_act(curr_frame)
        pixels = cur.B * cur.H * cur.W
        mv_est = self.opticFlow.run(cur, ref)
        mv_coarse, y_lik, z_lik = self.mv_codec.run_forward([cur, mv_est, ref], x_hat_dtype=F32)
        warped, warp_fea, predict = self._predict(ref, mv_coarse, sm_fea, self._feature_in(feature))
        bpp_mv = self._bpp([y_lik, z_lik], pixels)
        predict_fea = self.FeatureExtractor.run(predict)
        cur_fea = self.FeatureExtractor.run(cur)
        res = ops.eltwise(N.EW_SUB, cur_fea, predict_fea)
        res_coarse, y_lik, z_lik = self.res_codec.run_forward([ref, res, predict])
        bpp_res = self._bpp([y_lik, z_lik], pixels)
        feat, recon = self._reconstruct(ref, predict_fea, res_coarse, warp_fea)
        return cur, warped, predict, feat, recon, bpp_res, bpp_mv

    @torch.no_grad()
    def forward_msssim(self, ref_frame, curr_frame, sm_fea, feature=None):
        """video_model.py:96-135 -> (recon, feature, msssim, warp_msssim, mc_msssim, bpp_res, bpp_mv, bpp): the estimate
        path with MS-SSIM (data_range 1.0; deepsvc_amd.metrics.ms_ssim, a HIP kernel per scale) in place of the MSEs."""
        from .metrics import ms_ssim
        cur, warped, predict, feat, recon, bpp_res, bpp_mv = self._estimate(ref_frame, curr_frame, sm_fea, feature)
        return (recon.nchw(), feat.nchw(), ms_ssim(recon, cur, 1.0), ms_ssim(warped, cur, 1.0), ms_ssim(predict, cur, 1.0),
                bpp_res, bpp_mv, bpp_mv + bpp_res)

    # ------------------------------------------------------------------------------------------
    # real bitstream path
    # ------------------------------------------------------------------------------------------
    # Frame segments (include/dsvc.h: dsvc_pframe_segment_run): a P-frame is five host-synchronisation-free launch sequences -- E1
    # (motion estimation, motion analysis, encoder scan), E2 (motion synthesis, prediction, feature extraction, residual analysis,
    # encoder scan), D1 (motion synthesis, prediction, feature extraction), DR (residual synthesis, on the side stream) and D2
    # (residual refinement, enhancement, clamps) -- with the host range coder between them.  Each is recorded on its first execution
    # and replayed by ONE C call afterwards (ops.run_segment / ops.FrameProgram); every tensor that crosses a segment boundary is
    # passed in as an external, so a segment never depends on where its inputs were produced.
    def _segments(self, ref, *others):
        """(program cache, key prefix) for this call, or (None, None) when the call must run launch by launch (parity tests tap
        intermediate tensors).  Programs hold pointers to packed weights: any weight edit drops them."""
        if self.__dict__.get("debug_taps") is not None or not ops.FRAME_PROGRAMS[0]:
            return None, None
        ps = self.__dict__.get("_fp_params")
        if ps is None:
            ps = self.__dict__["_fp_params"] = list(self.parameters())
        # E1 / E2 inline the codecs' encoder scans (codec._step and its epoch check are bypassed inside a recording), so they freeze
        # pointers to gaussian_conditional.device_table() and the entropy bottleneck's parameter blob: a table update (update(),
        # update_scale_table(), load_state_dict) leaves the parameter versions alone and must drop the recordings too.  The table
        # generation is not moved by weight packing, so sampling it here -- before any recording of this call -- is race-free.
        fp = (sum(p._version for p in ps),) + tuple(m._table_gen for c in (self.mv_codec, self.res_codec)
                                                     for m in (c.entropy_bottleneck, c.gaussian_conditional))
        cache = self.__dict__.setdefault("_frame_programs", {})
        if self.__dict__.get("_fp_seen") != fp:
            cache.clear()
            self.__dict__["_fp_seen"] = fp
            # E1 / E2a / E2 / D1 / D2 run one after the other on the caller's stream and share one arena; DR runs beside D1 on the side stream
            self.__dict__["_arenas"] = (ops.Arena(), ops.Arena())
        sig = lambda t: (tuple(t.buf.shape), t.c0, t.C, t.code) if isinstance(t, Act) else \
            (None if t is None else (tuple(t.shape), str(t.dtype), tuple(t.stride())))
        return cache, (str(ref.device), ops.RT.mode, ops.RT.compute, sig(ref)) + tuple(sig(t) for t in others)

    @torch.no_grad()
    def compress(self, ref_frame, curr_frame, sm_fea, feature=None):
        """video_model.py:137-151 -> (mv_out_enc, res_out_enc), each {"strings": [[y],[z...]], "shape": (h,w)}."""
        _stamp("compress")
        ref, cur = to_act(ref_frame), to_act(curr_frame)
        self._maybe_capture_graphs(ref)
        feat = self._feature_in(feature)
        cache, key = self._segments(ref, cur, sm_fea, feat)
        ws_mv = self.mv_codec._plan(ref.B, ref.H // 16, ref.W // 16, ref.device)
        ws_res = self.res_codec._plan(ref.B, ref.H // 16, ref.W // 16, ref.device)
        opt = [] if feat is None else [feat]

        def e1(ref, cur):
            mv_est = self.opticFlow.run(cur, ref)
            self.mv_codec.enc_gpu([cur, mv_est, ref], ws_mv)
            return [], []

        def e2a():
            # (the reference entropy-decodes its own mv stream here, video_model.py:141: y_hat is already on the device)
            return [], [self.mv_codec._g_s(ws_mv.y_hat, None, F32)]

        def e2(ref, cur, sm, mv_coarse, *f):
            _, _, predict = self._predict(ref, mv_coarse, sm, f[0] if f else None)
            predict_fea = self.FeatureExtractor.run(predict)
            cur_fea = self.FeatureExtractor.run(cur)
            res = ops.eltwise(N.EW_SUB, cur_fea, predict_fea)
            self.res_codec.enc_gpu([ref, res, predict], ws_res)
            return [], []

        arena = self.__dict__.get("_arenas", (None, None))[0]
        ops.run_segment(cache, ("E1", key), [ref, cur], e1, arena)
        # the mv streams are range-coded on a worker thread while the GPU runs prediction + the residual analysis
        mv_pending = self.mv_codec.enc_finish(ws_mv, defer=True)
        _, (mv_coarse,) = ops.run_segment(cache, ("E2a", key), [], e2a, arena)
        # from here on nothing of the encoder touches the motion codec's scan workspace: the decoder's motion chain of this frame may
        # start on its own stream while the rest of the encoder still runs (decompress())
        done = self.__dict__["_mv_ws_free"] = torch.cuda.Event()
        done.record()
        ops.run_segment(cache, ("E2", key), [ref, cur, sm_fea, mv_coarse] + opt, e2, arena)
        res_pending = self.res_codec.enc_finish(ws_res, defer=True)
        # both are dicts that join their coder thread on first access: a caller that forwards them to decompress()
        # without looking inside overlaps the residual range coding with the decoder's motion path
        return mv_pending, res_pending

    @torch.no_grad()
    def decompress(self, ref_frame, mv_out_enc, res_out_enc, sm_fea, feature):
        """video_model.py:153-167 -> (feature, recon, warped, predict), the three frames clamped to [0,1]."""
        _stamp("decompress")
        ref = to_act(ref_frame)
        dev = ref.device
        self._maybe_capture_graphs(ref)
        feat = self._feature_in(feature)
        cache, key = self._segments(ref, sm_fea, feat)
        opt = [] if feat is None else [feat]
        arenas = self.__dict__.get("_arenas", (None, None))
        taps = self.__dict__.get("debug_taps")
        # The residual codec's decode chain (host range decoder + tiny scan kernels, 5 sync points) does not depend on
        # the motion path until reconstruction: it runs on a worker thread / side stream so its idle gaps overlap the
        # motion decode chain and the MC / feature-extraction kernels of the main stream.
        main = torch.cuda.current_stream(dev)
        # [r5] GOP lanes (use_side_stream = False): the chain runs on the lane's OWN stream (still from the worker thread, so the host decode
        # overlaps this thread's launches): with several lanes the other lanes' kernels fill the gaps anyway, and every extra stream costs --
        # HIP maps streams onto 4 hardware queues; one stream per lane instead of two: 4 lanes +2 %, 6 lanes +6 % (35.4-36.4 vs 33.7-33.9 fps)
        use_side = bool(getattr(self, "use_side_stream", True)) and not _env.get("DSVC_NO_SIDE_STREAM")
        side = main if not use_side else self.__dict__.setdefault("_side_streams", {}).setdefault(str(dev), torch.cuda.Stream(dev))
        side.wait_stream(main)
        res_done = torch.cuda.Event()

        def decode_residual():
            strings, shape = res_out_enc["strings"], res_out_enc["shape"]   # joins the encoder thread if still coding
            with torch.cuda.stream(side):
                ws = self.res_codec.dec_chain(strings, shape, dev)
                _, (out,) = ops.run_segment(cache, ("DR", key), [], lambda: ([], [self.res_codec._g_s(ws.y_hat)]), arenas[1])
                res_done.record(side)
            return out

        # GOP lanes (one stream per lane): the residual chain is issued by the lane thread itself once D1 is queued -- same rate as from a worker
        # thread (36.3-36.7 vs 36.4-36.7 fps at six lanes, three rounds), one host thread fewer per lane; DSVC_RES_WORKER=1 brings the worker back
        inline_res = (not use_side) and not _env.get("DSVC_RES_WORKER")
        res_future = None if inline_res else _DECODE_POOL.submit(decode_residual)
        # The motion decode chain (host range decoder <-> tiny scan kernels, one sync per level) depends on the mv STRINGS only, not on
        # anything the GPU is still doing for the encoder: with MV_STREAM it runs on its own high-priority stream, so when compress() of
        # this frame has just been queued (test_video.py:342,349 calls them back to back) part of its latency hides under the encoder's
        # remaining kernels (tiny kernels only get CU slots at the boundaries of the large ones, so the chain advances one launch per
        # large kernel: measured 7.9 ms for the chain beside E2 against ~4.5 ms alone).
        use_mvs = MV_STREAM[0] if MV_STREAM[0] is not None else bool(getattr(self, "single_stream", False))
        mvs = main if not use_mvs else self.__dict__.setdefault("_mv_streams", {}).get(str(dev))
        if mvs is None:
            mvs = self.__dict__["_mv_streams"][str(dev)] = torch.cuda.Stream(dev, priority=-1)
        free = self.__dict__.get("_mv_ws_free")
        with torch.cuda.stream(mvs):
            if free is not None:
                mvs.wait_event(free)                # the encoder has finished reading the shared scan workspace (y_hat)
            else:
                mvs.wait_stream(main)
            mv_strings = mv_out_enc["strings"]
            _stamp("mv strings joined")
            ws_mv = self.mv_codec.dec_chain(mv_strings, mv_out_enc["shape"], dev)
            _stamp("mv chain done")
        main.wait_stream(mvs)

        def d1(ref, sm, *f):
            mv_coarse = self.mv_codec._g_s(ws_mv.y_hat, None, F32)
            warped, warp_fea, predict = self._predict(ref, mv_coarse, sm, f[0] if f else None)
            predict_fea = self.FeatureExtractor.run(predict)
            if taps is not None:
                taps.update(mv_coarse=mv_coarse.nchw().float().clone())
            return [], [warped, warp_fea, predict, predict_fea]

        def d2(ref, warped, warp_fea, predict, predict_fea, res_coarse):
            fea, recon = self._reconstruct(ref, predict_fea, res_coarse, warp_fea)
            if taps is not None:      # parity tests look at the tensors in front of the clamps (a dict set by the caller; never set in production)
                taps.update(recon=recon.nchw().clone(), warped=warped.nchw().clone(), predict=predict.nchw().clone(),
                            warp_fea=warp_fea.nchw().float().clone(), predict_fea=predict_fea.nchw().float().clone(),
                            res_coarse=res_coarse.nchw().float().clone())
            clamp = lambda a: ops.eltwise(N.EW_CLAMP01, a)
            return [fea, clamp(recon), clamp(warped), clamp(predict)], []

        _, (warped, warp_fea, predict, predict_fea) = ops.run_segment(cache, ("D1", key), [ref, sm_fea] + opt, d1, arenas[0])
        _stamp("D1 queued")
        res_coarse = decode_residual() if inline_res else res_future.result()
        _stamp("res chain done")
        main.wait_event(res_done)
        res_coarse.buf.record_stream(main)
        (fea, recon_c, warped_c, predict_c), _ = ops.run_segment(cache, ("D2", key), [ref, warped, warp_fea, predict, predict_fea, res_coarse], d2, arenas[0])
        # `feature` is only ever handed back to the next frame: return it zero-copy in its storage layout/dtype
        return fea.nchw_view(), recon_c.nchw(), warped_c.nchw(), predict_c.nchw()

    # ------------------------------------------------------------------------------------------
    # bookkeeping (same contracts as the reference)
    # ------------------------------------------------------------------------------------------
    def aux_loss(self):
        return sum(m.loss() for m in self.modules() if isinstance(m, EntropyBottleneck))

    def mv_aux_loss(self):
        return sum(m.loss() for m in self.mv_codec.modules() if isinstance(m, EntropyBottleneck))

    def res_aux_loss(self):
        return sum(m.loss() for m in self.res_codec.modules() if isinstance(m, EntropyBottleneck))

    def set_coder(self, name):
        """Range coder of both codecs' y strings: "rans" (the reference's format, default) or "rans_ms" (GPU-decodable multi-stream
        coder, a labelled non-reference mode: image_model.ChannelSplitICIP2020ResB.coder)."""
        assert name in ("rans", "rans_ms")
        self.mv_codec.coder = self.res_codec.coder = name

    def update(self, force=False):
        updated = self.mv_codec.update(force=force)
        updated |= self.res_codec.update(force=force)
        return updated

    def load_state_dict(self, state_dict, strict=True):
        """video_model.py:184-191: the two codecs first resize their table buffers from their own sub-dicts."""
        state_dict = rename_legacy_keys(state_dict)
        for name in ("mv_codec", "res_codec"):
            sub = {k[len(name) + 1:]: v for k, v in state_dict.items() if k.startswith(name + ".")}
            getattr(self, name).load_state_dict(sub, strict=strict)
        return super().load_state_dict(state_dict, strict=strict)
