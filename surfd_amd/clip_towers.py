"""CLIP ViT-B/32 towers for the text / image / sketch conditioned drivers — the front end that turns a
prompt or a picture into the 512-d context of the denoiser.

  ClipTowers.encode_image(image [B,3,224,224])  <- CLIP/clip/model.py:206-240, 340-341  (VisionTransformer)
  ClipTowers.encode_text(tokens [B,77])         <- CLIP/clip/model.py:343-356           (causal Transformer, EOT pooling)
  SimpleTokenizer(bpe_path).tokenize(texts)     <- CLIP/clip/simple_tokenizer.py, clip.py:205-245 (byte-level BPE, truncate=True)

In the reference the text tower runs INSIDE MDM.forward, i.e. once per denoising step (models/mdm.py:86-97: 1000 times
per sample); here it runs once, before the loop, and its output enters the fused loop as `y['context']`.

Functional over a state_dict in OpenAI's checkpoint layout (visual.conv1.weight, visual.transformer.resblocks.N.attn.in_proj_weight,
token_embedding.weight, text_projection, ...): widths, depths and head counts are read off the tensor shapes, so any ViT-style
CLIP loads.  Plain torch ops (library GEMMs on the GPU): outside the hot path, evaluated once per request.  No weights
ship with this repository (none are available offline); the arithmetic is pinned against the reference's own model class
with seeded weights (tests/golden/g16_clip_towers.npz).
"""
from __future__ import annotations

import gzip
import html
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F
from torch import Tensor


def _ln(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    return F.layer_norm(x.float(), (x.shape[-1],), w.float(), b.float(), 1e-5).to(x.dtype)


class ClipTowers:
    def __init__(self, state_dict: Dict[str, Tensor]):
        sd = {k: v for k, v in state_dict.items() if isinstance(v, Tensor)}
        if "visual.proj" not in sd or "text_projection" not in sd:
            raise ValueError("ClipTowers needs a ViT-style CLIP state_dict (visual.proj, text_projection, ...)")
        self.sd = sd
        self.vision_width = sd["visual.conv1.weight"].shape[0]
        self.patch = sd["visual.conv1.weight"].shape[-1]
        self.vision_layers = len({k.split(".")[3] for k in sd if k.startswith("visual.transformer.resblocks.")})
        self.vision_heads = self.vision_width // 64
        self.text_width = sd["ln_final.weight"].shape[0]
        self.text_layers = len({k.split(".")[2] for k in sd if k.startswith("transformer.resblocks.")})
        self.text_heads = self.text_width // 64
        self.context_length = sd["positional_embedding"].shape[0]
        self.embed_dim = sd["text_projection"].shape[1]

    @classmethod
    def from_file(cls, path: str) -> "ClipTowers":
        """A plain state_dict file, a {'state_dict': ...} checkpoint, or OpenAI's TorchScript archive (ViT-B-32.pt)."""
        try:
            obj = torch.jit.load(path, map_location="cpu").state_dict()
        except RuntimeError:
            obj = torch.load(path, map_location="cpu")
            obj = obj.get("state_dict", obj) if isinstance(obj, dict) else obj.state_dict()
        return cls(obj)

    def to(self, device) -> "ClipTowers":
        self.sd = {k: v.to(device) for k, v in self.sd.items()}
        return self

    # one pre-norm residual block: x + MHA(LN1 x), then x + MLP(LN2 x) with QuickGELU
    def _block(self, x: Tensor, p: str, heads: int, mask: Optional[Tensor]) -> Tensor:
        sd = self.sd
        B, T, W = x.shape
        h = _ln(x, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"])
        qkv = F.linear(h, sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"]).view(B, T, 3, heads, W // heads)
        q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))                      # [B, heads, T, d]
        att = (q * (W // heads) ** -0.5) @ k.transpose(-1, -2)
        if mask is not None:
            att = att + mask
        o = (att.softmax(dim=-1) @ v).transpose(1, 2).reshape(B, T, W)
        x = x + F.linear(o, sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"])
        h = _ln(x, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"])
        h = F.linear(h, sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"])
        h = h * torch.sigmoid(1.702 * h)
        return x + F.linear(h, sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"])

    @torch.no_grad()
    def encode_image(self, image: Tensor) -> Tensor:
        sd = self.sd
        w = sd["visual.conv1.weight"]
        x = F.conv2d(image.to(w.dtype), w, stride=self.patch)                           # patch embedding, no bias
        x = x.flatten(2).transpose(1, 2)                                                # [B, grid^2, width]
        cls_tok = sd["visual.class_embedding"].to(x.dtype).expand(x.shape[0], 1, -1)
        x = torch.cat([cls_tok, x], dim=1) + sd["visual.positional_embedding"].to(x.dtype)
        x = _ln(x, sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"])
        for i in range(self.vision_layers):
            x = self._block(x, f"visual.transformer.resblocks.{i}.", self.vision_heads, None)
        x = _ln(x[:, 0], sd["visual.ln_post.weight"], sd["visual.ln_post.bias"])
        return x @ sd["visual.proj"]

    @torch.no_grad()
    def encode_text(self, tokens: Tensor) -> Tensor:
        sd = self.sd
        x = sd["token_embedding.weight"][tokens] + sd["positional_embedding"][: tokens.shape[1]]
        T = tokens.shape[1]
        mask = torch.full((T, T), float("-inf"), device=x.device, dtype=x.dtype).triu_(1)
        for i in range(self.text_layers):
            x = self._block(x, f"transformer.resblocks.{i}.", self.text_heads, mask)
        x = _ln(x, sd["ln_final.weight"], sd["ln_final.bias"])
        eot = tokens.argmax(dim=-1)                                                     # the end-of-text id is the largest in the vocabulary
        return x[torch.arange(x.shape[0], device=x.device), eot] @ sd["text_projection"]


def _byte_alphabet() -> Dict[int, str]:
    """The reversible byte -> printable-character map of GPT-2 style byte-level BPE: the 188 printable latin-1 bytes map to
    themselves, the other 68 (in increasing byte order) to code points 256, 257, ...  The dict's ORDER — printable bytes
    first — is the order of the first 256 vocabulary entries."""
    keep = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAD)) + list(range(0xAE, 0x100))
    table = {b: chr(b) for b in keep}
    for n, b in enumerate(b for b in range(256) if b not in table):
        table[b] = chr(256 + n)
    return table


class SimpleTokenizer:
    """CLIP's lower-cased byte-level BPE.  `bpe_path`: the merges file shipped with CLIP (bpe_simple_vocab_16e6.txt.gz — not
    part of this repository).  ftfy is not available here: text is cleaned with html.unescape + whitespace collapsing only
    (identical for plain ASCII prompts)."""

    def __init__(self, bpe_path: Optional[str]):
        import regex
        if not bpe_path:
            raise ValueError("SimpleTokenizer needs CLIP's BPE merges file (bpe_simple_vocab_16e6.txt.gz): pass --bpe_path")
        lines = gzip.open(bpe_path).read().decode("utf-8").split("\n")
        merges = [tuple(m.split()) for m in lines[1:49152 - 256 - 2 + 1]]
        alphabet = list(_byte_alphabet().values())
        vocab = alphabet + [c + "</w>" for c in alphabet] + ["".join(m) for m in merges] + ["<|startoftext|>", "<|endoftext|>"]
        self.encoder = {tok: i for i, tok in enumerate(vocab)}
        self.rank = {m: i for i, m in enumerate(merges)}
        self.bytes = _byte_alphabet()
        self.cache = {"<|startoftext|>": "<|startoftext|>", "<|endoftext|>": "<|endoftext|>"}
        self.pattern = regex.compile(r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""", regex.IGNORECASE)
        self.sot, self.eot = self.encoder["<|startoftext|>"], self.encoder["<|endoftext|>"]

    def _bpe(self, token: str) -> str:
        if token in self.cache:
            return self.cache[token]
        word = list(token[:-1]) + [token[-1] + "</w>"]
        while len(word) > 1:
            pairs = {(word[i], word[i + 1]) for i in range(len(word) - 1)}
            best = min(pairs, key=lambda p: self.rank.get(p, float("inf")))
            if best not in self.rank:
                break
            merged, i = [], 0
            while i < len(word):
                if i + 1 < len(word) and (word[i], word[i + 1]) == best:
                    merged.append(word[i] + word[i + 1])
                    i += 2
                else:
                    merged.append(word[i])
                    i += 1
            word = merged
        out = " ".join(word)
        self.cache[token] = out
        return out

    def encode(self, text: str) -> List[int]:
        import regex
        text = regex.sub(r"\s+", " ", html.unescape(html.unescape(text)).strip()).strip().lower()
        ids: List[int] = []
        for tok in self.pattern.findall(text):
            tok = "".join(self.bytes[b] for b in tok.encode("utf-8"))
            ids.extend(self.encoder[piece] for piece in self._bpe(tok).split(" "))
        return ids

    def tokenize(self, texts: Sequence[str], context_length: int = 77) -> Tensor:
        """[len(texts), context_length] int64, start / end markers added, over-long prompts cut with the end marker kept
        (clip.tokenize(..., truncate=True), models/mdm.py:88)."""
        if isinstance(texts, str):
            texts = [texts]
        out = torch.zeros(len(texts), context_length, dtype=torch.long)
        for i, t in enumerate(texts):
            ids = [self.sot] + self.encode(t) + [self.eot]
            if len(ids) > context_length:
                ids = ids[:context_length]
                ids[-1] = self.eot
            out[i, : len(ids)] = torch.tensor(ids)
        return out
