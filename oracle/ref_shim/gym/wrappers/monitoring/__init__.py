from gym.wrappers.monitoring import video_recorder
