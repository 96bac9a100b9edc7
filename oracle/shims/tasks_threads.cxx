// ORACLE / TEST INFRASTRUCTURE ONLY - never linked into libetx_hip.so.
//
// etx::TaskScheduler on a plain std::thread pool.
// The reference implements this interface (sources/etx/render/host/tasks.hxx:23-46) on enkiTS in
// render/host/tasks.cxx:47-160; that TU cannot be used here because its pimpl buffer (296 B, tasks.hxx:45) is
// smaller than its Impl on libstdc++. The interface and the observable behaviour are kept:
//   schedule() returns immediately, completed() polls, wait() blocks and invalidates the handle,
//   execute() = schedule+wait, ranges are split in chunks handed to workers with a thread id < max_thread_count().
#include <etx/render/host/tasks.hxx>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include <cstdlib>

namespace etx {

namespace {

struct FuncTask : public Task {
  std::function<void(uint32_t, uint32_t, uint32_t)> fn;
  void execute_range(uint32_t b, uint32_t e, uint32_t t) override {
    fn(b, e, t);
  }
};

struct Job {
  Task* task = nullptr;
  std::unique_ptr<FuncTask> owned;
  uint32_t range = 0;
  uint32_t chunk = 1;
  std::atomic<uint32_t> next{0};
  std::atomic<uint32_t> done{0};
  bool in_use = false;
};

struct Pool {
  std::vector<std::thread> workers;
  std::vector<std::shared_ptr<Job>> jobs;  // a slot gets a fresh Job object on reuse, so late workers only ever see finished jobs
  std::deque<uint32_t> queue;
  std::mutex mtx;
  std::condition_variable cv_work;
  std::condition_variable cv_done;
  bool quit = false;
  uint32_t thread_count = 1;

  Pool() {
    uint32_t hw = std::thread::hardware_concurrency();
    if (const char* e = getenv("ETX_ORACLE_THREADS")) {
      hw = uint32_t(atoi(e));
    }
    thread_count = hw > 0 ? hw : 1u;
    for (uint32_t i = 0; i < thread_count; ++i) {
      workers.emplace_back([this, i]() {
        run(i + 1u);
      });
    }
  }

  ~Pool() {
    {
      std::unique_lock<std::mutex> l(mtx);
      quit = true;
    }
    cv_work.notify_all();
    for (auto& w : workers)
      w.join();
  }

  bool work_on(Job& j, uint32_t tid) {
    bool did = false;
    for (;;) {
      uint32_t b = j.next.fetch_add(j.chunk);
      if (b >= j.range)
        break;
      uint32_t e = b + j.chunk < j.range ? b + j.chunk : j.range;
      j.task->execute_range(b, e, tid);
      did = true;
      if (j.done.fetch_add(e - b) + (e - b) >= j.range) {
        std::unique_lock<std::mutex> l(mtx);
        cv_done.notify_all();
      }
    }
    return did;
  }

  void run(uint32_t tid) {
    for (;;) {
      std::shared_ptr<Job> job;
      {
        std::unique_lock<std::mutex> l(mtx);
        cv_work.wait(l, [this]() {
          return quit || (queue.empty() == false);
        });
        if (quit)
          return;
        job = jobs[queue.front()];
        if (job->next.load() >= job->range) {
          queue.pop_front();
          continue;
        }
      }
      work_on(*job, tid);
    }
  }

  uint32_t add(uint32_t range, Task* t, std::unique_ptr<FuncTask> owned) {
    std::unique_lock<std::mutex> l(mtx);
    uint32_t idx = ~0u;
    for (uint32_t i = 0; i < jobs.size(); ++i) {
      if (jobs[i]->in_use == false) {
        idx = i;
        break;
      }
    }
    if (idx == ~0u) {
      idx = uint32_t(jobs.size());
      jobs.emplace_back();
    }
    jobs[idx] = std::make_shared<Job>();
    Job& j = *jobs[idx];
    j.in_use = true;
    j.task = t;
    j.owned = std::move(owned);
    j.range = range;
    uint32_t parts = thread_count * 32u;
    j.chunk = range / parts > 0 ? range / parts : 1u;
    j.next = 0;
    j.done = 0;
    if (range > 0) {
      queue.push_back(idx);
      cv_work.notify_all();
    }
    return idx;
  }
};

}  // namespace

struct TaskSchedulerImpl {
  std::unique_ptr<Pool> pool = std::make_unique<Pool>();
};

TaskScheduler::TaskScheduler() {
  ETX_PIMPL_INIT(TaskScheduler);
}

TaskScheduler::~TaskScheduler() {
  ETX_PIMPL_CLEANUP(TaskScheduler);
}

uint32_t TaskScheduler::max_thread_count() {
  return _private->pool->thread_count + 2u;
}

void TaskScheduler::register_thread() {
}

Task::Handle TaskScheduler::schedule(uint32_t range, Task* t) {
  return {_private->pool->add(range, t, nullptr)};
}

Task::Handle TaskScheduler::schedule(uint32_t range, std::function<void(uint32_t, uint32_t, uint32_t)> func) {
  auto owned = std::make_unique<FuncTask>();
  owned->fn = func;
  Task* t = owned.get();
  return {_private->pool->add(range, t, std::move(owned))};
}

void TaskScheduler::execute(uint32_t range, Task* t) {
  auto h = schedule(range, t);
  wait(h);
}

void TaskScheduler::execute(uint32_t range, std::function<void(uint32_t, uint32_t, uint32_t)> func) {
  auto h = schedule(range, func);
  wait(h);
}

void TaskScheduler::execute_linear(uint32_t range, std::function<void(uint32_t, uint32_t, uint32_t)> func) {
  func(0u, range, 0u);
}

bool TaskScheduler::completed(Task::Handle handle) {
  if (handle.data == Task::InvalidHandle)
    return true;
  std::unique_lock<std::mutex> l(_private->pool->mtx);
  Job& j = *_private->pool->jobs[handle.data];
  return j.done.load() >= j.range;
}

void TaskScheduler::wait(Task::Handle& handle) {
  if (handle.data == Task::InvalidHandle)
    return;
  Pool& p = *_private->pool;
  std::shared_ptr<Job> jp;
  {
    std::unique_lock<std::mutex> l(p.mtx);
    jp = p.jobs[handle.data];
  }
  Job& j = *jp;
  // the waiting thread helps (thread id 0), like enkiTS' WaitforTask
  p.work_on(j, 0u);
  {
    std::unique_lock<std::mutex> l(p.mtx);
    p.cv_done.wait(l, [&j]() {
      return j.done.load() >= j.range;
    });
    for (auto it = p.queue.begin(); it != p.queue.end();) {
      it = (*it == handle.data) ? p.queue.erase(it) : it + 1;
    }
    j.owned.reset();
    j.task = nullptr;
    j.in_use = false;
  }
  handle.data = Task::InvalidHandle;
}

void TaskScheduler::restart(Task::Handle handle) {
  if (handle.data == Task::InvalidHandle)
    return;
  Pool& p = *_private->pool;
  Job& j = *p.jobs[handle.data];
  p.work_on(j, 0u);
  std::unique_lock<std::mutex> l(p.mtx);
  p.cv_done.wait(l, [&j]() {
    return j.done.load() >= j.range;
  });
  j.next = 0;
  j.done = 0;
  p.queue.push_back(handle.data);
  p.cv_work.notify_all();
}

}  // namespace etx
