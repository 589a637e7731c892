class VideoRecorder: pass
