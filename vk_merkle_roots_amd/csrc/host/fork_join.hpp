// fork_join.hpp -- a tiny persistent fork-join pool for the parallel packer: Run(n, f) calls
// f(0..n-1), task 0 on the caller, the others on parked worker threads.
#pragma once
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace vkmr {

class ForkJoin {
public:
    explicit ForkJoin(unsigned workers) : m_stop(false), m_generation(0), m_pending(0), m_tasks(0)
    {
        for (unsigned i = 0; i < workers; ++i) m_threads.emplace_back([this, i] { Loop(i + 1); });
    }
    ~ForkJoin()
    {
        {
            std::lock_guard<std::mutex> g(m_mu);
            m_stop = true;
            ++m_generation;
        }
        m_wake.notify_all();
        for (auto& t : m_threads) t.join();
    }
    unsigned Width() const { return (unsigned)m_threads.size() + 1u; }

    // f(i) for i in [0, n), n <= Width(); returns when all are done.
    void Run(unsigned n, const std::function<void(unsigned)>& f)
    {
        if (n == 0) return;
        {
            std::lock_guard<std::mutex> g(m_mu);
            m_fn = &f;
            m_tasks = n;
            m_pending = n - 1;
            ++m_generation;
        }
        m_wake.notify_all();
        f(0);
        std::unique_lock<std::mutex> g(m_mu);
        m_done.wait(g, [this] { return m_pending == 0; });
        m_fn = nullptr;
    }

private:
    void Loop(unsigned id)
    {
        unsigned long seen = 0;
        for (;;) {
            const std::function<void(unsigned)>* fn = nullptr;
            {
                std::unique_lock<std::mutex> g(m_mu);
                m_wake.wait(g, [&] { return m_generation != seen; });
                seen = m_generation;
                if (m_stop) return;
                if (id < m_tasks) fn = m_fn;
            }
            if (fn) {
                (*fn)(id);
                std::lock_guard<std::mutex> g(m_mu);
                if (--m_pending == 0) m_done.notify_one();
            }
        }
    }

    std::vector<std::thread> m_threads;
    std::mutex m_mu;
    std::condition_variable m_wake, m_done;
    bool m_stop;
    unsigned long m_generation;
    unsigned m_pending, m_tasks;
    const std::function<void(unsigned)>* m_fn = nullptr;
};

}  // namespace vkmr
