"""CPU oracle for the DeepSVC P-frame hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in `deepsvc_amd/` (the product) may import, call, link
or execute anything in this package: only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg use it, and there only as the checker / CPU baseline.

What it is
----------
A plain PyTorch-CPU (fp32) *restatement* of the reference algorithm, written functionally over
a ``state_dict`` whose keys are the reference's own parameter names:

* ``oracle.nets``    -- modules.py blocks (warp, SpyNet, res-blocks, SE, UNet, RefineNet,
                        Reconstruction, FeatureExtraction, InterLayerPrediction)
* ``oracle.entropy`` -- compressai 1.2.1 arithmetic the path relies on (EntropyBottleneck,
                        GaussianConditional, pmf_to_quantized_cdf, rANS coder)
* ``oracle.codec``   -- image_model.py:ChannelSplitICIP2020ResB and video_model.py:DeepSVC
                        (forward / compress / decompress)
* ``oracle/rans_ref.c`` -- C restatement of the rANS coder + pmf_to_quantized_cdf, built by
                        ``oracle/build.py`` into ``oracle/_build/`` (used for sizes where the
                        pure-Python coder is too slow).

Pinning status
--------------
* nets + codec tensor arithmetic: PINNED against the reference
  itself, imported in the build container with stand-ins for its absent third-party packages;
  golden vectors committed under ``tests/golden/`` together with the generating script
  ``tests/golden/make_golden.py``.
* entropy arithmetic (rows a16-a18: compressai 1.2.1 EntropyBottleneck / GaussianConditional /
  pmf_to_quantized_cdf / rans_interface.cpp over ryg_rans rans64.h, pinned in the reference's
  env.txt:12 but NOT vendored under /root/reference and not installed here):
  **parity unpinned** -- restated from the published algorithm; the
  reference holds no tests, golden vectors or fixtures for it.  Checked by properties only
  (CDF monotone / sums to 2**16, encode->decode identity incl. escape symbols, Python == C).
"""
