"""Voice packs: ``[510, 1, 256]`` style tables indexed by phoneme count (reference: kokoro/voice.py, pipeline.py:303)."""
import torch


def load_voice_tensor(path: str) -> torch.Tensor:
    """Loads a ``.safetensors`` voice file (tensor name ``voice``) or a PyTorch ``.pt`` pack."""
    if str(path).endswith(".safetensors"):
        from safetensors.torch import load_file

        return load_file(str(path))["voice"]
    return torch.load(str(path), map_location="cpu", weights_only=True)
