from .fully_control import FullySelfAttentionControlMask, MutualSelfAttentionControl  # noqa: F401
from .fully_control_utils import MutualAttentionBase, regiter_fully_attention_editor_diffusers  # noqa: F401
from .temporal_control import TemporalSelfAttentionControl  # noqa: F401
from .temporal_control_utils import TemporalAttentionBase, regiter_temporal_attention_editor_diffusers  # noqa: F401
