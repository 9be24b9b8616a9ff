#!/usr/bin/env python3
"""Weight interchange: the reference's trained weights -> the flat blob aipt_denoise_load_weights takes.

    python tools/export_weights.py cpp_autoencoder_3.pt weights.aiptw        # TorchScript archive (convert_to_torchscript.py:29-30)
    python tools/export_weights.py autoencoder_model_12_3.pt weights.aiptw   # training checkpoint {'net': state_dict} (train.py:109-112)
    aiptd scene.txt --weights weights.aiptw ...

Blob layout: ai_path_tracer_denoiser_amd/arch.py (header + 28 x {W, b, gamma, beta, running_mean, running_var} in
named_parameters()/named_buffers() order)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    if len(sys.argv) != 3:
        sys.exit(__doc__)
    from ai_path_tracer_denoiser_amd import arch
    blob = arch.blob_from_file(sys.argv[1])
    with open(sys.argv[2], "wb") as f:
        f.write(blob)
    print(f"{sys.argv[2]}: {len(blob)} bytes, {len(arch.layer_table())} conv+BN layers")


if __name__ == "__main__":
    main()
