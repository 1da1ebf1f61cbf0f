"""Drop-in for the reference's ``net.py`` on the inference path, running on MI355X HIP kernels.

Same public surface as ``/root/reference/net.py``: module-level ``vgg`` (53 children,
net.py:76-130) and ``decoder`` (38 children, net.py:33-74) with the reference's state-dict keys,
``Net(encoder, decoder, decoder_ph=None, distance_g=None)`` (net.py:133-260) and ``Distance_G()``
(net.py:266-327).  The children are parameter holders; the arithmetic is the fused plan of
``style_transfer_based_holographic_imaging_amd.engine`` (NHWC fp32 MFMA convolutions, fused
ReLU/pool epilogues).  ``Net.forward`` evaluates the training objective's forward values (no autograd: the HIP kernels have no
backward pass); ``Discriminator`` is out of scope.  Tensors must live on a ROCm device: there is no CPU fallback.
"""
import torch
import torch.nn as nn

from function import calc_mean_std
from style_transfer_based_holographic_imaging_amd import ops
from style_transfer_based_holographic_imaging_amd import torch_ops  # noqa: F401  (registers the holo:: operators with the dispatcher)
from style_transfer_based_holographic_imaging_amd.engine import HoloSequential


def _pad():
    return nn.ReflectionPad2d((1, 1, 1, 1))


def _conv_block(cin, cout, relu=True):
    layers = [_pad(), nn.Conv2d(cin, cout, (3, 3))]
    if relu:
        layers.append(nn.ReLU())
    return layers


def _up(c):
    return [nn.ConvTranspose2d(c, c, 2, stride=2, padding=0), nn.ReLU()]


def _pool():
    return nn.MaxPool2d((2, 2), (2, 2), (0, 0), ceil_mode=True)


def _build_decoder():
    # 512 ->512 ->256 ^ 256 x3 ->128 ^ 128 ->64 ^ 64 x2 -> 2   (^ = ConvTranspose2d k2 s2)
    spec = [(512, 512), (512, 256), "up256", (256, 256), (256, 256), (256, 256), (256, 128), "up128",
            (128, 128), (128, 64), "up64", (64, 64), (64, 64)]
    layers = []
    for item in spec:
        layers += _up(int(item[2:])) if isinstance(item, str) else _conv_block(*item)
    layers += _conv_block(64, 2, relu=False)
    return HoloSequential(*layers)


def _build_vgg():
    # VGG-19 with a 1->3 1x1 pre-conv; "P" = 2x2 max-pool with ceil_mode
    widths = [(3, 64), (64, 64), "P", (64, 128), (128, 128), "P", (128, 256), (256, 256), (256, 256), (256, 256),
              "P", (256, 512), (512, 512), (512, 512), (512, 512), "P", (512, 512), (512, 512), (512, 512),
              (512, 512)]
    layers = [nn.Conv2d(1, 3, (1, 1))]
    for item in widths:
        layers += [_pool()] if item == "P" else _conv_block(*item)
    return HoloSequential(*layers)


decoder = _build_decoder()
vgg = _build_vgg()


class Net(nn.Module):
    """net.py:133-260.  ``encoder`` is an ``nn.Sequential`` (e.g. ``vgg`` truncated to 31 children)."""

    def __init__(self, encoder, decoder, decoder_ph=None, distance_g=None):
        super().__init__()
        if isinstance(encoder, str):
            raise TypeError("Net(encoder=<path>) relies on an Encoder class the reference never defines; "
                            "pass the nn.Sequential encoder instead")
        enc_layers = list(encoder.children())
        bounds = [0, 4, 11, 18, 31] + ([44] if len(enc_layers) > 35 else [])
        self.n_layer = len(bounds) - 1
        for k in range(self.n_layer):
            setattr(self, f"enc_{k + 1}", HoloSequential(*enc_layers[bounds[k]:bounds[k + 1]]))
        self.decoder = decoder if isinstance(decoder, HoloSequential) else HoloSequential(*decoder.children())
        self.decoder_ph = decoder_ph
        self.distance_g = distance_g
        self.mse_loss = nn.MSELoss()
        for k in range(self.n_layer):
            for param in getattr(self, f"enc_{k + 1}").parameters():
                param.requires_grad = False

    def set_compute_dtype(self, dtype):
        """Additive: torch.float32 (default, exact fp32 MFMA), torch.bfloat16 for the conv stack (BASELINE config 3), or
        "f32s": fp32 activations and weights kept as two fp16 halves (22-23 significant bits) so that the conv stack runs
        on the fp16 matrix cores with fp32 accumulation — measured closer to float64 than the fp32 path, at more than
        twice its speed, but NOT bit-compatible with fp32 arithmetic, hence opt-in.  Inputs, amplitude/phase outputs,
        statistics and the distance head stay fp32."""
        for k in range(self.n_layer):
            getattr(self, f"enc_{k + 1}").set_compute_dtype(dtype)
        self.decoder.set_compute_dtype(dtype)
        return self

    def set_conv_algo(self, algo):
        """Additive: "auto" (default), "direct" or "winograd" for the fp32 3x3 layers (engine.HoloSequential.set_conv_algo)."""
        for k in range(self.n_layer):
            getattr(self, f"enc_{k + 1}").set_conv_algo(algo)
        self.decoder.set_conv_algo(algo)
        return self

    @staticmethod
    def _public(feat):
        """Feature maps leave the module as fp32 (the optional "f32s" compute mode keeps them as split halves inside)."""
        if feat.dtype == ops.F32S:
            return ops.as_nchw_view(ops.cast(ops.to_nhwc(feat), torch.float32))
        return feat

    def encode_with_intermediate(self, input):
        results = [input]
        for i in range(self.n_layer):
            results.append(getattr(self, "enc_{:d}".format(i + 1))(results[-1]))
        return [self._public(r) for r in results[1:]]

    def _encode(self, input, intensity=False):
        for i in range(self.n_layer):
            stage = getattr(self, "enc_{:d}".format(i + 1))
            input = stage(input, intensity=True) if (intensity and i == 0) else stage(input)
        return input

    def encode(self, input):
        return self._public(self._encode(input))

    def calc_content_loss(self, input, target):
        """net.py:186-189"""
        assert (input.size() == target.size())
        assert (target.requires_grad is False)
        return self.mse_loss(input, target)

    def calc_style_loss(self, input, target):
        """net.py:191-197: MSE of the per-channel means plus MSE of the per-channel standard deviations"""
        assert (input.size() == target.size())
        assert (target.requires_grad is False)
        input_mean, input_std = calc_mean_std(input)
        target_mean, target_std = calc_mean_std(target)
        return self.mse_loss(input_mean, target_mean) + self.mse_loss(input_std, target_std)

    def forward(self, content, style, alpha=1.0, field_retrieval=False, unkonwn_distance=False):
        """net.py:199-226: the training objective.  Same arguments (sic: ``unkonwn_distance``) and the same 2 / 5 / 7 return
        values; every network pass runs on the HIP kernels (encoder on the style image and on the generated amplitude,
        image-style AdaIN of function.py:15-23, two decoder passes, the distance head on both statistics); the scalar MSE
        reductions are ``nn.MSELoss`` as in the reference.  With autograd recording (``torch.enable_grad``, parameters that
        require gradients) the results carry a graph whose every node is a HIP kernel pair (forward / backward,
        ``style_transfer_based_holographic_imaging_amd/autograd.py``): ``loss.backward()`` fills the gradients of the decoder and
        of the distance head, like the reference's aten graph (fp32; ``Distance_G`` in eval mode has no Dropout, in training mode it
        draws its masks from torch's generator and applies them in the kernels).
        Under ``torch.no_grad()`` it evaluates the values only."""
        assert 0 <= alpha <= 1
        with torch.no_grad():       # the encoder is frozen and the inputs carry no gradient: the AdaIN target is a constant (net.py:188,194)
            style_feats = self.encode_with_intermediate(style)
            content_feat = self.encode(content)
            # adain(content_feat, style_feats[-1]) and the alpha blend, in one launch (function.py:15-23, net.py:203-204)
            x = ops.to_nhwc(content_feat)
            s_mean, s_std = ops.channel_mean_std(ops.to_nhwc(style_feats[-1]))
            c_mean, c_std = ops.channel_mean_std(x)
            t = ops.as_nchw_view(ops.adain_apply(x, c_mean, c_std, s_mean, s_std, float(alpha)))
        # from here on autograd records (when enabled): two decoder passes, the encoder on the generated amplitude, the losses.
        # Every step is an autograd Function on the HIP kernels (style_transfer_based_holographic_imaging_amd/autograd.py).
        style_re = self.decoder(style_feats[-1])[:, :1]
        g_t = self.decoder(t)                 # content diffraction pattern -> style diffraction pattern
        g_t_phase = g_t[:, 1:]
        g_t = g_t[:, :1]
        g_t_feats = self.encode_with_intermediate(g_t)
        loss_c = self.calc_content_loss(g_t_feats[-1], t)
        loss_s = self.calc_style_loss(g_t_feats[0], style_feats[0])
        for i in range(1, self.n_layer):
            loss_s += self.calc_style_loss(g_t_feats[i], style_feats[i])
        if field_retrieval:
            if unkonwn_distance:
                d_content = self.distance_g(calc_mean_std(content_feat))
                d_style = self.distance_g(calc_mean_std(style_feats[-1]))
                return loss_c, loss_s, g_t, g_t_phase, style_re, d_content, d_style
            return loss_c, loss_s, g_t, g_t_phase, style_re
        return loss_c, loss_s

    def field_retrieval(self, content, style_feats, alpha=1.0, unknown_distance=False, intensity=False):
        """net.py:228-260.  ``intensity`` (additive): ``content`` is the hologram intensity and the first layer's loader takes the
        square root — the entry loop's ``torch.sqrt(content_holo)`` (test_field_retrieval_mnist.py:119) without its own pass."""
        assert 0 <= alpha <= 1
        content_feat = self._encode(content, intensity)          # logical NCHW, channels-last memory
        feat_nhwc = ops.to_nhwc(content_feat)                    # free view
        b = content.shape[0]
        style_feats = style_feats.to(device=content_feat.device, dtype=torch.float32)
        c = feat_nhwc.shape[-1]
        if style_feats.shape[0] > 2:   # multiple style method: first half means, second half stds
            half = style_feats.shape[0] // 2
            style_mean, style_std = style_feats[:half].reshape(half, c), style_feats[half:].reshape(-1, c)
            if half != b:
                raise RuntimeError(f"multiple-style vector holds {half} styles for a batch of {b}")
        else:                          # single style broadcast over the batch
            style_mean, style_std = style_feats[:1].reshape(1, c), style_feats[1:].reshape(1, c)
        t, mean, std = ops.adain_fused(feat_nhwc, style_mean, style_std, float(alpha))   # statistics + AdaIN, one launch
        g_t = self.decoder.run_nhwc(t)                           # (B,2,S,S) planes
        g_t_phase = g_t[:, 1:]
        g_t = g_t[:, :1]
        if unknown_distance:
            # B == 1: the reference repeats the row to dodge InstanceNorm1d's batch-1 check and keeps row 0
            # (net.py:255-256); the per-row normalisation here gives that same value directly.
            return g_t, g_t_phase, self.distance_g((mean.view(b, c, 1, 1), std.view(b, c, 1, 1)))
        return g_t, g_t_phase


class Distance_G(nn.Module):
    """net.py:266-327: MLP 1024 -> 1024 -> 1024 -> 512 -> 1 on [mean, std] statistics."""

    def __init__(self):
        super().__init__()
        in_fc = 512 * 2
        widths = [(in_fc, in_fc), (in_fc, in_fc), (in_fc, in_fc // 2)]
        for k, (fin, fout) in enumerate(widths, start=1):
            setattr(self, f"l{k}", nn.Linear(in_features=fin, out_features=fout, bias=True))
            setattr(self, f"b{k}", nn.InstanceNorm1d(num_features=fout))
            setattr(self, f"d{k}", nn.Dropout(p=0.5))
        self.relu = nn.ReLU()
        self.out = nn.Linear(in_features=in_fc // 2, out_features=1)
        self.initialize_weights()

    def forward(self, m_s):
        m, s = m_s
        assert m.shape == s.shape
        b, c = m.shape[:2]
        recording = torch.is_grad_enabled() and (m.requires_grad or s.requires_grad or any(p.requires_grad for p in self.parameters()))
        if self.training or recording:
            from style_transfer_based_holographic_imaging_amd import autograd as _ag
            keep, scale = None, 1.0
            if self.training and self.d1.p > 0:
                # training mode: nn.Dropout on each Linear's output (net.py:298-304).  The 0/1 masks are drawn by torch's generator
                # (no arithmetic); the kernels apply them forward and backward.
                keep = [torch.empty((b, lin.out_features), dtype=torch.float32, device=m.device).bernoulli_(1.0 - d.p)
                        for lin, d in ((self.l1, self.d1), (self.l2, self.d2), (self.l3, self.d3))]
                scale = 1.0 / (1.0 - self.d1.p)
            return _ag.DistanceHead.apply(m.reshape(b, c), s.reshape(b, c), self.l1.weight, self.l1.bias, self.l2.weight, self.l2.bias,
                                          self.l3.weight, self.l3.bias, self.out.weight, self.out.bias, keep, scale)
        sd = {f"{n}.{p}": getattr(getattr(self, n), p) for n in ("l1", "l2", "l3", "out") for p in ("weight", "bias")}
        return ops.distance_head(m.reshape(b, c).float(), s.reshape(b, c).float(), sd)

    def initialize_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
