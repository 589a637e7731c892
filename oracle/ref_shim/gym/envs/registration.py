def register(id, entry_point=None, max_episode_steps=None, **kw):
    import gym
    gym._registry[id] = dict(entry_point=entry_point, max_episode_steps=max_episode_steps)
