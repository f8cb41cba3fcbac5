from .pipeline_motion_editor import MotionEditorPipeline, MotionEditorPipelineOutput  # noqa: F401
