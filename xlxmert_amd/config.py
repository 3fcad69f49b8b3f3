"""Model-side configuration of the path: the LxmertConfig fields the reference reads (HFcfg:72-101) plus the
ad-hoc attributes its wrappers set (`num_clusters`, `clustering`; ref lxrt/modeling.py:64-65)."""
from dataclasses import asdict, dataclass


@dataclass
class XLxmertConfig:
    vocab_size: int = 30522
    hidden_size: int = 768
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    hidden_act: str = "gelu"                    # exact-erf GELU (ACT2FN["gelu"])
    hidden_dropout_prob: float = 0.1
    attention_probs_dropout_prob: float = 0.1
    max_position_embeddings: int = 512
    type_vocab_size: int = 2
    initializer_range: float = 0.02
    l_layers: int = 9
    x_layers: int = 5
    r_layers: int = 5
    visual_feat_dim: int = 2048
    visual_pos_dim: int = 4
    num_clusters: int = 10000                   # --n_centroids (ref param.py:167)
    layer_norm_eps: float = 1e-12
    visual_obj_loss: bool = True
    visual_feat_loss: bool = True               # the model code computes both
    task_obj_predict: bool = True
    task_mask_lm: bool = True                   # `cls` heads are built when task_mask_lm or task_matched (ref lxrt/modeling.py:85-86)
    task_matched: bool = True
    task_qa: bool = False                       # --taskQA is off in scripts/pretrain.bash; True builds `answer_head` (ref :89-90)
    num_qa_labels: int = 9500                   # LxmertConfig default
    use_return_dict: bool = True

    @property
    def clustering(self):
        return self.num_clusters > 0

    @property
    def n_centroids(self):
        return self.num_clusters

    @property
    def head_dim(self):
        return self.hidden_size // self.num_attention_heads

    def to_dict(self):
        return asdict(self)

    def __post_init__(self):
        if self.hidden_size % self.num_attention_heads != 0:
            raise ValueError(f"The hidden size ({self.hidden_size}) is not a multiple of the number of attention "
                             f"heads ({self.num_attention_heads})")
        if self.hidden_act != "gelu":
            raise ValueError("only the exact-erf 'gelu' activation is implemented (what the reference uses)")
