class EzPickle:
    def __init__(self,*a,**k): pass
