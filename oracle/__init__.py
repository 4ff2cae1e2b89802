"""CPU oracle for the StableTTS CFM-decoder hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``stabletts_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg use it, and there only as the checker.

The oracle is a plain fp32 PyTorch-CPU *restatement* of
  models/estimator.py, models/diffusion_transformer.py and the arithmetic of
  models/flow_matching.py (+ torchdiffeq's fixed-grid solvers)
of the reference at /root/reference.  It is pinned against the real reference
modules by ``oracle/make_golden.py`` (run in the build container, where the
reference is mounted) -> ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
re-checks the restatement against those vectors on every run.

``text_encoder_forward`` restates models/text_encoder.py and is pinned the same way
(``oracle/make_golden_text_encoder.py`` -> ``tests/golden/text_encoder_outputs.npz``).

Parity status: PINNED for the estimator (Decoder.forward and every sub-module)
against outputs of the reference's own modules.  The ODE solver is torchdiffeq
(un-vendored, unpinned in requirements.txt:17, absent offline): its fixed-grid
euler / midpoint / rk4(3/8) rules are restated from the published algorithm and
are "parity unpinned" against torchdiffeq itself.
"""
from .weights import (DecoderConfig, make_state_dict, make_cfg_params,  # noqa: F401
                      TextEncoderConfig, make_text_encoder_state_dict)
from .estimator_oracle import (  # noqa: F401
    decoder_forward, cfm_forward, cfg_wrapper, compute_loss, odeint_fixed,
    sinusoidal_pos_emb, rope, attention, ffn, dit_block, linspace_f32, odeint_dopri5,
    dit_conv_block, text_encoder_forward, odeint_adaptive, ADAPTIVE_TABLEAUS, odeint_implicit_adams, adams_coefficients,
)
from .inputs import make_inputs  # noqa: F401
