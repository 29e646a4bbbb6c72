def colored(text, *a, **k):
    return text
